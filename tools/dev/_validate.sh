cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05v
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/r05v/gpu_tests.txt
timeout 900 python tools/gpu_wsweep_full.py > gpurun_out/r05v/wsweep_full.txt 2>&1; tail -2 gpurun_out/r05v/wsweep_full.txt
timeout 600 python tools/gpu_share_check.py > gpurun_out/r05v/share_check.txt 2>&1; tail -2 gpurun_out/r05v/share_check.txt
timeout 600 python tools/gpu_wsweep_constrained.py > gpurun_out/r05v/wsweep_constrained.txt 2>&1; tail -2 gpurun_out/r05v/wsweep_constrained.txt

set -x
mkdir -p gpurun_out/r04
cd /root/repo
export TMPDIR=/tmp
./tools/micro/issue_rates > gpurun_out/r04/issue_rates.json 2> gpurun_out/r04/issue_rates.err
tail -c 600 gpurun_out/r04/issue_rates.json
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /root/repo/gpurun_out/r04/ir_pmc1 -- /root/repo/tools/micro/issue_rates > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/r04/ir_pmc2 -- /root/repo/tools/micro/issue_rates > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /root/repo/gpurun_out/r04/ir_pmc3 -- /root/repo/tools/micro/issue_rates > /dev/null 2>&1)
python tools/gpu_mfe_cmp.py 262144 120 > gpurun_out/r04/occ_cmp.txt 2>&1
cat gpurun_out/r04/occ_cmp.txt
python -m pytest tests/test_gpu_paths.py -x -q > gpurun_out/r04/paths_tests.log 2>&1
tail -15 gpurun_out/r04/paths_tests.log
python bench.py > gpurun_out/r04/bench1.json 2> gpurun_out/r04/bench1.err
tail -c 3000 gpurun_out/r04/bench1.json; tail -5 gpurun_out/r04/bench1.err

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05p
rm -f gpurun_out/r05p/pf_stamps.txt
mkdir -p /tmp/abl_only && cp tools/abl_pfstamps.so /tmp/abl_only/
SF_STAMP_OUT=gpurun_out/r05p/pf_stamps.txt timeout 600 python tools/gpu_pf_scan_cmp.py 120 1 > gpurun_out/r05p/pf_scan_cmp.txt 2>&1
tail -3 gpurun_out/r05p/pf_scan_cmp.txt
python tools/dev/pf_stamp_report.py gpurun_out/r05p/pf_stamps.txt | tee gpurun_out/r05p/pf_stamp_report.txt

#!/bin/bash
# Partition-function A/B in one gpurun call: tools/gpu_pf_scan_cmp.py (the product library against every tools/abl_*.so build variant,
# tools/dev/abl.py) at the widths of $PF_AB_WIDTHS; with a cycle-stamped build parked as tools/dev/abl_pfstamps.so.keep, its per-team
# profile (tools/dev/pf_stamp_report.py) as well; then the GPU suite's partition-function tests.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/pf_cmp.sh'
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/pf_cmp
for W in ${PF_AB_WIDTHS:-120 100 60}; do timeout 600 python tools/gpu_pf_scan_cmp.py $W 1 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/pf_cmp/pf_ab.txt
if [ -f tools/dev/abl_pfstamps.so.keep ]; then
  rm -f tools/abl_*.so; cp tools/dev/abl_pfstamps.so.keep tools/abl_pfstamps.so
  rm -f gpurun_out/pf_cmp/pf_stamps.txt
  SF_STAMP_OUT=gpurun_out/pf_cmp/pf_stamps.txt timeout 600 python tools/gpu_pf_scan_cmp.py 120 1 2>&1 | grep -v amdgpu.ids | tail -1
  python tools/dev/pf_stamp_report.py gpurun_out/pf_cmp/pf_stamps.txt | tee gpurun_out/pf_cmp/pf_stamp_report.txt
fi
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "partition or share" 2>&1 | tail -3

#!/bin/bash
# per-window PMC counters of the LDS partition-function kernel (rocprofv3 counter passes; run on the GPU box)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/pfpmc2
N=${N:-8192}
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64"; do
  d=$R/gpurun_out/pfpmc2/$(echo $c | tr " " "_" | cut -c1-40)
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/gpu_pf_only.py $N 120 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$R/gpurun_out/pfpmc2/*/*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "sf_pf_lds" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    print({k: "%.3g" % (v/$N) for k,v in acc.items()})
PY

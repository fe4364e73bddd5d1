#!/bin/bash
# per-fold PMC counters of the batched MFE kernel (rocprofv3 counter passes; run on the GPU box)
# usage: tools/pmc_mfe.sh [W (120)] [n folds (131072)]
W=${1:-120}; N=${2:-131072}
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/pkpmc
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  d=$R/gpurun_out/pkpmc/$(echo $c | tr " " "_" | cut -c1-40)
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $R/tools/gpu_mfe_only.py $N $W > /dev/null 2>&1
done
python3 - <<PY
import csv, glob
tot = {}
for f in sorted(glob.glob("$R/gpurun_out/pkpmc/*/*/*counter_collection.csv")):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "sf_mfe_" in r["Kernel_Name"] and "full" not in r["Kernel_Name"] and int(r["Grid_Size"]) >= 256*128:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    tot.update(acc)
    print({k: "%.4g" % (v/$N) for k,v in acc.items()})
if "SQ_WAVE_CYCLES" in tot:
    wpf = 4 if $W <= 128 else 8
    T = tot["SQ_WAVE_CYCLES"] / wpf
    print("W=$W per fold: VALU busy %.3f  LDS busy %.3f  lanes/VALU inst %.1f  waves waiting %.3f  (busy = unit-active quad-cycles x SIMD share / fold residency)" % (
        tot["SQ_ACTIVE_INST_VALU"] * (16 // wpf // 1) / 4 / T * (wpf / 4) if False else tot["SQ_ACTIVE_INST_VALU"] / T * (16 / wpf) / 4,
        tot["SQ_LDS_IDX_ACTIVE"] / T * (16 / wpf) / 4, tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_ACTIVE_INST_VALU"], tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"]))
PY

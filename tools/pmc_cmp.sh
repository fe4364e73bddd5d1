#!/bin/bash
# counters of the batched MFE kernel for the product library and every tools/abl_*.so (rocprofv3 counter passes)
# usage: bash tools/pmc_cmp.sh "CTR1 CTR2 ..." ["CTR3 ..." ...]   (each argument = one pass)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_cmp; rm -rf $OUT; mkdir -p $OUT
W=${W:-120}; N=${N:-131072}
for lib in "" $(cd $R && ls tools/abl_*.so 2>/dev/null); do
  tag=$(basename "${lib:-product}" .so)
  k=0
  for c in "$@"; do
    k=$((k+1))
    SCANFOLD_LIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$tag/p$k -- python3 $R/tools/gpu_mfe_only.py $N $W > /dev/null 2>&1
  done
done
python3 - <<PY
import csv, glob, os
for d in sorted(glob.glob("$OUT/*")):
    acc = {}
    for f in glob.glob(d + "/*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "sf_mfe_fast" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 1024 * 128:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    print(os.path.basename(d), {k: "%.4g" % (v / $N) for k, v in sorted(acc.items())})
PY

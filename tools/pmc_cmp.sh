#!/bin/bash
# counters of the batched MFE kernel for the product library and every tools/abl_*.so (rocprofv3 counter passes)
# usage: bash tools/pmc_cmp.sh "CTR1 CTR2 ..." ["CTR3 ..." ...]   (each argument = one pass)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_cmp; rm -rf $OUT; mkdir -p $OUT
W=${W:-120}; N=${N:-131072}; T=${T:-90}
for lib in "" $(cd $R && ls tools/abl_*.so 2>/dev/null); do
  tag=$(basename "${lib:-product}" .so)
  k=0
  for c in "$@"; do
    k=$((k+1))
    SCANFOLD_LIB=$lib timeout $T rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$tag/p$k -- python3 $R/tools/gpu_mfe_only.py $N $W > $OUT/$tag.p$k.log 2>&1
    echo "$tag pass $k rc=$?" >> $OUT/progress.txt
  done
done
python3 - <<PY
import csv, glob, os
for d in sorted(glob.glob("$OUT/*/")):
    acc = {}
    for f in glob.glob(d + "/*/*/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if "sf_mfe_fast" in r["Kernel_Name"]]
        last = max(int(r["Dispatch_Id"]) for r in rows) if rows else -1   # the timed launch (the warm-up launch comes first)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    print(os.path.basename(d.rstrip("/")), {k: "%.4g" % (v / $N) for k, v in sorted(acc.items())})
PY

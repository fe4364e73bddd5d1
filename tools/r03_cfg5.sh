#!/bin/bash
# Round-3 measurement set for BASELINE config 5 (W=200, r=1000, partition function) on ONE GPU: bench.py --config cfg5, the
# same under rocprofv3 --kernel-trace --stats, FETCH_SIZE / WRITE_SIZE passes, SQ counter passes of the W=200 MFE kernel
# alone, summary -> gpurun_out/$V/cfg5_mfe_counters.json (copy to profiles/r03/).
R=$GRAFT_REPO_ROOT; V=${V:-r03_cfg5}; O=$R/gpurun_out/$V
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
python3 $R/bench.py --config cfg5 > $O/cfg5_bench.json 2> $O/cfg5_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config cfg5 --no-cpu-baseline --no-live-counters > $O/cfg5_bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/tools/gpu_mfe_only.py 262144 200 > /dev/null 2>&1
done
k=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
  k=$((k+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/sq_$k -- python3 $R/tools/gpu_mfe_only.py 65536 200 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json
O = "$O"
def pmc_sum(pattern, kernel, min_grid=0):
    acc = {}
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and int(r["Grid_Size"]) >= min_grid:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return acc
# traffic passes: the W = 200 MFE kernel alone on 262 144 random 200-mers (the launch of >= 512 workgroups; the warm-up launch is smaller)
fetch = pmc_sum(O + "/pmc_FETCH_SIZE/*/*counter_collection.csv", "sf_mfe_fast_kernel", 512 * 512)
write = pmc_sum(O + "/pmc_WRITE_SIZE/*/*counter_collection.csv", "sf_mfe_fast_kernel", 512 * 512)
NT = 262144.0
sq = {}
for k in (1, 2, 3):
    sq.update(pmc_sum(O + "/sq_%d/*/*counter_collection.csv" % k, "sf_mfe_fast_kernel", 512 * 256))
N = 65536.0
per_fold = {k: v / N for k, v in sq.items()}
out = {"source": "tools/r03_cfg5.sh on MI355X; rocprofv3 --pmc passes, one counter group per run",
       "hbm_bytes_per_fold": (2 * fetch.get("FETCH_SIZE", 0) + write.get("WRITE_SIZE", 0)) * 1024 / NT,
       "hbm_note": "sf_mfe_fast_kernel<256,200> on 262 144 random 200-mers: (2 x FETCH_SIZE (gfx950 reports half of a read) + WRITE_SIZE) KB -> bytes, per fold; L2 <-> fabric traffic, Infinity-Cache hits included; bench.py multiplies by the folds of one launch",
       "per_fold_counters_W200": per_fold}
if per_fold.get("SQ_WAVE_CYCLES"):
    wc = per_fold["SQ_WAVE_CYCLES"]
    out["secondary"] = {
        "valu_busy": 4 * per_fold.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        "lanes_active_of_64": per_fold.get("SQ_THREAD_CYCLES_VALU", 0) / max(per_fold.get("SQ_ACTIVE_INST_VALU", 1), 1),
        "lds_busy": 4 * per_fold.get("SQ_LDS_IDX_ACTIVE", 0) / wc,
        "waves_parked": per_fold.get("SQ_WAIT_ANY", 0) / wc,
        "valu_insts_per_fold": per_fold.get("SQ_INSTS_VALU"), "lds_insts_per_fold": per_fold.get("SQ_INSTS_LDS"),
        "salu_insts_per_fold": per_fold.get("SQ_INSTS_SALU"),
        "definition": "SQ counters of sf_mfe_fast_kernel<256,200> on 65 536 random 200-mers, per fold; eight waves per fold, two folds "
                      "per CU, so a SIMD hosts four waves: busy = 4 x unit-active quad-cycles / wave quad-cycles of a fold"}
json.dump(out, open(O + "/cfg5_mfe_counters.json", "w"), indent=1)
print(json.dumps(out)[:1200])
PY
cat $O/cfg5_bench.json
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs head -6

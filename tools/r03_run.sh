#!/bin/bash
# Round-3 measurement set (run on the GPU box through gpurun): bench.py, the same under rocprofv3 --kernel-trace --stats,
# FETCH_SIZE / WRITE_SIZE passes of the same command, SQ counter passes of the MFE kernel alone, and the summary JSON
# bench.py reads (profiles/r03/mfe_counters.json).  Output: gpurun_out/$V/ ; copy what should be judged to profiles/r03/.
R=$GRAFT_REPO_ROOT; V=${V:-r03_final}; O=$R/gpurun_out/$V
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --no-cpu-baseline --no-live-counters > $O/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-live-counters > /dev/null 2>&1
done
k=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
  k=$((k+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/sq_$k -- python3 $R/tools/gpu_mfe_only.py 131072 120 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, json, os
O = "$O"
def pmc_sum(pattern, kernel, min_grid=0, per=None):
    acc = {}
    n = 0
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and int(r["Grid_Size"]) >= min_grid:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return acc
def pmc_max(pattern, kernel):  # the largest dispatch = the one full cfg3 launch of the timed step (the e2e leg launches chunks)
    best = {}
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                best[r["Counter_Name"]] = max(best.get(r["Counter_Name"], 0.0), float(r["Counter_Value"]))
    return best
fetch = pmc_max(O + "/pmc_FETCH_SIZE/*/*counter_collection.csv", "sf_mfe_fast_kernel")
write = pmc_max(O + "/pmc_WRITE_SIZE/*/*counter_collection.csv", "sf_mfe_fast_kernel")
sq = {}
for k in (1, 2, 3):
    sq.update(pmc_sum(O + "/sq_%d/*/*counter_collection.csv" % k, "sf_mfe_fast_kernel", 1024 * 128))
N = 131072.0
per_fold = {k: v / N for k, v in sq.items()}
out = {"source": "tools/r03_run.sh on MI355X; rocprofv3 --pmc passes, one counter group per run (superseded by tools/r04_run.sh: counters of the timed workload, no warm-up launch)",
       "hbm_bytes_per_launch": (2 * fetch.get("FETCH_SIZE", 0) + write.get("WRITE_SIZE", 0)) * 1024,
       "hbm_note": "MFE kernel launch of one cfg3 step (3 017 981 folds, the largest dispatch of the pass): 2 x FETCH_SIZE (gfx950 reports half of a read) + WRITE_SIZE, KB -> bytes; L2 <-> fabric traffic, Infinity-Cache hits included (profiles/r03/mfe_scratch_traffic.txt)",
       "fetch_size_kb": fetch.get("FETCH_SIZE"), "write_size_kb": write.get("WRITE_SIZE"),
       "per_fold_counters_W120": per_fold}
if per_fold.get("SQ_WAVE_CYCLES"):
    wc = per_fold["SQ_WAVE_CYCLES"]
    out["secondary"] = {
        "valu_busy": 4 * per_fold.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        "lanes_active_of_64": per_fold.get("SQ_THREAD_CYCLES_VALU", 0) / max(per_fold.get("SQ_ACTIVE_INST_VALU", 1), 1),
        "lds_busy": 4 * per_fold.get("SQ_LDS_IDX_ACTIVE", 0) / wc,
        "waves_parked": per_fold.get("SQ_WAIT_ANY", 0) / wc,
        "valu_insts_per_fold": per_fold.get("SQ_INSTS_VALU"), "lds_insts_per_fold": per_fold.get("SQ_INSTS_LDS"),
        "salu_insts_per_fold": per_fold.get("SQ_INSTS_SALU"),
        "definition": "SQ counters of sf_mfe_fast_kernel<128,120> on 131 072 random 120-mers, per fold; a SIMD hosts four waves "
                      "(one of each of the CU's four folds): busy = 4 x unit-active quad-cycles / wave quad-cycles of a fold"}
json.dump(out, open(O + "/mfe_counters.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
cat $O/bench.json
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs head -8

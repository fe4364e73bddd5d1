#!/bin/bash
# Round-4 measurement set (run on the GPU box through gpurun):  bash tools/r04_run.sh [cfg3|cfg5] [extra bench flags]
#   bench.py --config C (live counters on), the same under rocprofv3 --kernel-trace --stats, and the counter passes — every one
#   of them on tools/gpu_scan_only.py C: ONE scan of the timed workload with no warm-up launch, so that the counters are the timed
#   kernel's (round 3 took them on random 120-mers and counted the warm-up launch in).  Summary -> gpurun_out/$V/*mfe_counters.json
#   (what bench.py --no-live-counters reads once copied to profiles/r04/).
R=$GRAFT_REPO_ROOT; C=${1:-cfg3}; shift; V=${V:-r04_$C}; O=$R/gpurun_out/$V
P=""; [ "$C" = cfg3 ] || P="${C}_"
WIN=""; [ "$C" = cfg5 ] && WIN="--windows 2000"
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
python3 $R/bench.py --config $C "$@" > $O/${P}bench.json 2> $O/${P}bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config $C --no-cpu-baseline --no-live-counters "$@" > $O/${P}bench_under_rocprof.json 2>/dev/null
k=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  k=$((k+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$k -- python3 $R/tools/gpu_scan_only.py $C $WIN > $O/pmc_$k.json 2>/dev/null
done
python3 - <<PY
import csv, glob, json, sys
sys.path.insert(0, "$R")
import bench
O, C, P = "$O", "$C", "$P"
wl = bench.WORKLOADS[C]
acc, disp, dur = {}, {}, []
for k in range(1, 7):
    for f in glob.glob(O + "/pmc_%d/*/*counter_collection.csv" % k):
        for r in csv.DictReader(open(f)):
            if "sf_mfe_fast_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                disp.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
    for f in glob.glob(O + "/pmc_%d/*/*kernel_trace.csv" % k):
        for r in csv.DictReader(open(f)):
            if "sf_mfe_fast_kernel" in r["Kernel_Name"] and k == 1:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
run = json.load(open(O + "/pmc_1.json"))
folds, launches = run["folds"], run["launches"]
per_fold = {c: v / folds for c, v in acc.items()}
b = json.load(open(O + "/" + P + "bench.json"))
sec = {"lanes_active_of_64": per_fold["SQ_THREAD_CYCLES_VALU"] / per_fold["SQ_ACTIVE_INST_VALU"],
       "waves_parked": per_fold["SQ_WAIT_ANY"] / per_fold["SQ_WAVE_CYCLES"],
       "valu_insts_per_fold": per_fold["SQ_INSTS_VALU"], "lds_insts_per_fold": per_fold["SQ_INSTS_LDS"],
       "salu_insts_per_fold": per_fold["SQ_INSTS_SALU"], "lds_idx_active_per_fold": per_fold["SQ_LDS_IDX_ACTIVE"],
       "lds_bank_conflict_per_fold": per_fold.get("SQ_LDS_BANK_CONFLICT")}
# the fractions for THIS run's launches: the child scan's own launch time (kernel trace of the first pass; counters slow nothing down)
ms = sum(dur) / max(len(dur), 1)
sec.update(bench.calibrated_unit_fractions(sec, wl["W"], folds / launches, ms))
out = {"source": "tools/r04_run.sh %s on MI355X; rocprofv3 --pmc passes of tools/gpu_scan_only.py %s $WIN (one counter group per run; "
                 "every sf_mfe_fast_kernel dispatch of the scan, no warm-up launch)" % (C, C),
       "folds": folds, "launches": launches, "dispatches_seen": {c: len(s) for c, s in disp.items()},
       "launch_ms_under_the_profiler": ms,
       "hbm_bytes_per_fold": (2 * per_fold["FETCH_SIZE"] + per_fold["WRITE_SIZE"]) * 1024,
       "hbm_note": "(2 x FETCH_SIZE (gfx950 reports half of a read) + WRITE_SIZE) KB -> bytes per fold; L2 <-> fabric traffic, "
                   "Infinity-Cache hits included; bench.py multiplies by the folds of one launch",
       "per_fold_counters": per_fold, "secondary": sec,
       "bench_value_of_the_same_run": b.get("value"), "bench_avg_launch_ms": b["roofline"]["avg_launch_ms"]}
if C == "cfg3":
    out["hbm_bytes_per_launch"] = out["hbm_bytes_per_fold"] * folds / launches
json.dump(out, open(O + "/" + P + "mfe_counters.json", "w"), indent=1)
print(json.dumps(out)[:1800])
PY
cat $O/${P}bench.json | head -c 1500; echo
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs head -8

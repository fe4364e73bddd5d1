#!/bin/bash
# One round's measurement set, parameterised by the round tag (replaces r02_run.sh / r03_run.sh / r03_cfg5.sh / r04_run.sh /
# r04_final.sh / r04_collect.sh / final_run.sh).  Run on the GPU box through gpurun; `collect` runs in the build container.
#
#   bash tools/run_round.sh measure RR cfg3|cfg5 [extra bench flags]   bench.py --config C (live counters on), the same under
#        rocprofv3 --kernel-trace --stats, and six counter passes of tools/gpu_scan_only.py C — ONE scan of the timed workload, no
#        warm-up launch, so that the counters are the timed kernel's — summarised in gpurun_out/RR_final/C/[C_]mfe_counters.json
#   bash tools/run_round.sh final RR       the GPU test suite, `measure` for cfg3 and cfg5, the secondary workloads of SURVEY 8d
#        (mononucleotide shuffles; viral-like input), the one-GPU shard-step estimate -> gpurun_out/RR_final/
#   bash tools/run_round.sh sweeps RR      every width 16..256 against the oracle (plain, constrained), shared against stand-alone inside
#        tables, the parity campaign, the partition-function time inside a scan at W = 120 / 100 / 60 -> gpurun_out/RR_final/sweeps/
#   bash tools/run_round.sh collect RR     copies what `final` left under gpurun_out/RR_final into profiles/RR/ (the tracked copies;
#        bench.py --no-live-counters reads profiles/RR/*mfe_counters.json)
# RR = r05, r06, ...
MODE=$1; RR=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S=$R/gpurun_out/${RR}_final

measure() {
  local C=${1:-cfg3}; shift
  local O=$S/$C P="" WIN=""
  [ "$C" = cfg3 ] || P="${C}_"
  [ "$C" = cfg5 ] && WIN="--windows 2000"
  cd /tmp && export TMPDIR=/tmp
  mkdir -p $O
  python3 $R/bench.py --config $C "$@" > $O/${P}bench.json 2> $O/${P}bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config $C --no-cpu-baseline --no-live-counters "$@" > $O/${P}bench_under_rocprof.json 2>/dev/null
  local k=0
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
ms = sum(dur) / max(len(dur), 1)  # the child scan's own launch time (kernel trace of the first pass)
sec.update(bench.calibrated_unit_fractions(sec, wl["W"], folds / launches, ms))
out = {"source": "tools/run_round.sh measure $RR %s on MI355X; rocprofv3 --pmc passes of tools/gpu_scan_only.py %s $WIN (one counter group per run; "
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
print(json.dumps(out)[:1500])
PY
  head -c 1200 $O/${P}bench.json; echo
  find $O/prof -name "*kernel_stats.csv" | head -1 | xargs head -6
}

final() {
  mkdir -p $S; cd $R
  python3 -m pytest tests -m gpu -x -q > $S/gpu_tests.log 2>&1; tail -6 $S/gpu_tests.log
  measure cfg3 > $S/cfg3.log 2>&1
  measure cfg5 > $S/cfg5.log 2>&1
  cd $R
  python3 bench.py --shuffle mono --no-cpu-baseline > $S/bench_mono.json 2> $S/bench_mono.err
  python3 bench.py --input viral --no-cpu-baseline > $S/bench_viral.json 2> $S/bench_viral.err
  python3 tools/gpu_shard_step.py cfg3 > $S/shard_step_cfg3.txt 2>&1
  python3 tools/gpu_shard_step.py cfg5 --windows 4096 --reps 1 > $S/shard_step_cfg5_first_4096_windows.txt 2>&1
  for f in cfg3/bench.json cfg5/cfg5_bench.json bench_mono.json bench_viral.json; do python3 -c "
import json
j=json.load(open('$S/$f')); print('$f', round(j['value'],1), j['verified_mismatches'], round(j['roofline']['avg_launch_ms'],1), j['config']['workload'])"; done
  tail -7 $S/shard_step_cfg3.txt
}

sweeps() {
  cd $R; mkdir -p $S/sweeps
  timeout 900 python3 tools/gpu_wsweep_full.py > $S/sweeps/wsweep_full.txt 2>&1; tail -1 $S/sweeps/wsweep_full.txt
  timeout 600 python3 tools/gpu_wsweep_constrained.py > $S/sweeps/wsweep_constrained.txt 2>&1; tail -1 $S/sweeps/wsweep_constrained.txt
  timeout 600 python3 tools/gpu_share_check.py > $S/sweeps/share_check.txt 2>&1; tail -1 $S/sweeps/share_check.txt
  timeout 900 python3 tools/gpu_parity_campaign.py > $S/sweeps/parity_campaign.txt 2>&1; tail -1 $S/sweeps/parity_campaign.txt
  for W in 120 100 60; do timeout 600 python3 tools/gpu_pf_scan_cmp.py $W 1 2>&1 | grep -v amdgpu.ids; done | tee $S/sweeps/pf_scan_times.txt
}

collect() {
  local D=$R/profiles/$RR; mkdir -p $D
  newest() { ls -t $1 2>/dev/null | head -1; }
  cp $S/cfg3/bench.json $D/final_bench.json
  cp $S/cfg3/bench_under_rocprof.json $D/final_bench_under_rocprof.json
  cp $S/cfg3/mfe_counters.json $D/mfe_counters.json
  cp "$(newest "$S/cfg3/prof/*/*kernel_stats.csv")" $D/final_bench_kernel_stats.csv
  for k in 1 2 3 4 5 6; do cp "$(newest "$S/cfg3/pmc_$k/*/*counter_collection.csv")" $D/final_pmc_pass$k.csv; done
  cp $S/cfg5/cfg5_bench.json $D/cfg5_bench.json
  cp $S/cfg5/cfg5_bench_under_rocprof.json $D/cfg5_bench_under_rocprof.json
  cp $S/cfg5/cfg5_mfe_counters.json $D/cfg5_mfe_counters.json
  cp "$(newest "$S/cfg5/prof/*/*kernel_stats.csv")" $D/cfg5_bench_kernel_stats.csv
  for k in 1 2 3 4 5 6; do cp "$(newest "$S/cfg5/pmc_$k/*/*counter_collection.csv")" $D/cfg5_pmc_pass$k.csv; done
  cp $S/bench_mono.json $D/secondary_bench_mono.json
  cp $S/bench_viral.json $D/secondary_bench_viral.json
  cp $S/gpu_tests.log $D/gpu_tests.log
  cp $S/shard_step_cfg3.txt $S/shard_step_cfg5_first_4096_windows.txt $D/ 2>/dev/null
  cp $S/sweeps/*.txt $D/ 2>/dev/null
  ls $D
}

case "$MODE" in
  measure) measure "$@" ;;
  final) final ;;
  sweeps) sweeps ;;
  collect) collect ;;
  *) echo "usage: bash tools/run_round.sh measure|final|sweeps|collect RR [cfg3|cfg5] [bench flags]"; exit 2 ;;
esac

#!/bin/bash
# Round-4 final measurement set: cfg3 and cfg5 (tools/r04_run.sh), SURVEY 8d's secondary workloads as bench lines
# (mononucleotide shuffles; the viral-like 60 % AU input with planted hairpins), the full GPU test suite.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_final; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
V=r04_final/cfg3 bash tools/r04_run.sh cfg3 > $O/cfg3.log 2>&1
V=r04_final/cfg5 bash tools/r04_run.sh cfg5 > $O/cfg5.log 2>&1
python3 bench.py --shuffle mono --no-cpu-baseline > $O/bench_mono.json 2> $O/bench_mono.err
python3 bench.py --input viral --no-cpu-baseline > $O/bench_viral.json 2> $O/bench_viral.err
for f in cfg3/bench.json cfg5/cfg5_bench.json bench_mono.json bench_viral.json; do python3 -c "
import json,sys
j=json.load(open('$O/$f')); print('$f', round(j['value'],1), j['verified_mismatches'], round(j['roofline']['avg_launch_ms'],1), j['config']['workload'])"; done

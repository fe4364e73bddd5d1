#!/bin/bash
# copies what tools/r04_final.sh left under gpurun_out/r04_final into profiles/r04 (newest file of each kind)
S=gpurun_out/r04_final; D=profiles/r04
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

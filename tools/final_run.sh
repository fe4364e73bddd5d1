set -x
R=$GRAFT_REPO_ROOT; V=${V:-v12}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$V
python3 $R/bench.py > $R/gpurun_out/$V/bench.json 2> $R/gpurun_out/$V/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$V/prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/$V/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/$V/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2>&1
done
cat $R/gpurun_out/$V/bench.json
find $R/gpurun_out/$V -name "*kernel_stats.csv" | head -2

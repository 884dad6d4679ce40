set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r04_s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
stats() { local name=$1; shift; rm -rf $OUT/stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/rocprof_$name.err || tail -5 $OUT/rocprof_$name.err
  find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \; ; rm -rf $OUT/stats; echo "stats $name done"; }
stats head --steps 20 --warmup 3 --no-cpu --no-workloads --headline-only --sustain 0
stats dense --only-workload dense
stats config4_doubled --only-workload config4_doubled
stats config5 --only-workload config5

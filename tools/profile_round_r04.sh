#!/usr/bin/env bash
# usage (on the GPU box): bash tools/profile_round_r04.sh <tag>     e.g. r04_a
# Everything the driver's bench line times, at HEAD, with the rocprofv3 summaries behind it -> gpurun_out/prof_<tag>/
# (copy what is to be judged into profiles/):
#   <tag>_bench.json                      the bench line (sub-records: dense, config4 native / doubled, config5)
#   <tag>_kernel_stats.csv                rocprofv3 --kernel-trace --stats of the headline's timed region
#   <tag>_<workload>_kernel_stats.csv     the same for every sub-record's workload (bench.py --only-workload)
#   <tag>_pmc_kpe_pc.json                 PMC passes of the extraction kernel on the headline stream (FETCH_SIZE, WRITE_SIZE,
#                                         SQ_INSTS_VALU ... ; separate passes, kernel trace only beside them)
#   <tag>_dense_pmc_kpe_pc_dense.json, <tag>_config4_native_pmc_kpe_pc.json, <tag>_blit_pmc_fgm_blit.json (the scroller's canvas blit: frames_per_s_with_blit)
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
python3 bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
echo "bench line done"
cd /tmp && export TMPDIR=/tmp
stats() { # name, args...
  local name=$1; shift
  rm -rf $OUT/stats
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py "$@" > $OUT/${name}_under_rocprof.json 2> $OUT/rocprof_$name.err || tail -5 $OUT/rocprof_$name.err
  find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/stats
  echo "stats $name done"
}
stats ${TAG} --steps 20 --warmup 3 --no-cpu --no-workloads --headline-only --sustain 0
for wl in dense config4_native config4_doubled config5; do
  stats ${TAG}_$wl --only-workload $wl
done
cd $R
pmc() { # out name, kernel substring, env..., -- script args
  local name=$1 kern=$2; shift 2
  ( export "$@" ; N=$PN DENS=$PD bash tools/pmc_kpe_r04.sh ${TAG}_$name ) > $OUT/pmc_$name.log 2>&1 || tail -5 $OUT/pmc_$name.log
  python3 tools/pmc_to_json.py gpurun_out/pmc_${TAG}_$name "$kern" $OUT/${TAG}_$name.json > /dev/null || true
  rm -rf gpurun_out/pmc_${TAG}_$name
  echo "pmc $name done"
}
PN="10000 200" PD="" pmc pmc_kpe_pc "kpe_pc<4" RMX_PC_DENSE=0
PN="10000 40" PD="" pmc dense_pmc_kpe_pc_dense "kpe_pc<8" RMX_PC_DENSE=1
PN="10000 200 640 480" PD="" pmc config4_native_pmc_kpe_pc "kpe_pc<4" RMX_PC_DENSE=0
bash tools/pmc_blit.sh ${TAG}_blit > $OUT/pmc_blit.log 2>&1 || tail -5 $OUT/pmc_blit.log
cp gpurun_out/pmc_${TAG}_blit/pmc_fgm_blit.json $OUT/${TAG}_blit_pmc_fgm_blit.json || true
rm -rf gpurun_out/pmc_${TAG}_blit
# the bit-plane blit kernel (fgm_bs.hip, behind RMX_BLIT_BS=1), for the record
( export RMX_BLIT_BS=1; bash tools/pmc_blit.sh ${TAG}_blit_bs ) > $OUT/pmc_blit_bs.log 2>&1 || tail -5 $OUT/pmc_blit_bs.log
cp gpurun_out/pmc_${TAG}_blit_bs/pmc_fgm_blit.json $OUT/${TAG}_blit_pmc_fgm_blit_bs.json || true
rm -rf gpurun_out/pmc_${TAG}_blit_bs
ls -la $OUT
head -c 1200 $OUT/${TAG}_bench.json

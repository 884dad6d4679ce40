#!/usr/bin/env bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>     e.g. r02_a
# bench line (with the other workloads as sub-records), rocprofv3 --kernel-trace --stats of the same bench
# command (without the sub-records), PMC passes of the extraction kernel -> gpurun_out/prof_<tag>/ (copy what is to be judged
# into profiles/).
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $R
python3 bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu --no-workloads --headline-only > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/rocprof.err || tail -5 $OUT/rocprof.err
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
cd $R
# the extraction kernel's two finishes (a context picks by density; forced here so each pass profiles one kernel)
RMX_BS_RING=1 bash tools/pmc_kpe.sh ${TAG}_ring > $OUT/pmc_ring.log 2>&1 || tail -5 $OUT/pmc_ring.log
python3 tools/pmc_to_json.py gpurun_out/pmc_${TAG}_ring "kpe_bs_ring(" $OUT/${TAG}_pmc_kpe_bs_ring.json > /dev/null
RMX_BS_RING=0 bash tools/pmc_kpe.sh ${TAG}_wg > $OUT/pmc_wg.log 2>&1 || tail -5 $OUT/pmc_wg.log
python3 tools/pmc_to_json.py gpurun_out/pmc_${TAG}_wg "kpe_bs(" $OUT/${TAG}_pmc_kpe_bs.json > /dev/null
rm -rf $OUT/stats gpurun_out/pmc_${TAG}_ring gpurun_out/pmc_${TAG}_wg
ls -la $OUT
head -c 1500 $OUT/${TAG}_bench.json

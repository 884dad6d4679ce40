#!/usr/bin/env bash
# kpe_mf time and L2-miss read traffic (FETCH_SIZE) against rows per strip (RMX_RPS); on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/rps
mkdir -p $OUT
for rps in ${RPS_LIST:-8 16 24 48}; do
  export RMX_RPS=$rps
  python3 $R/tools/perf_kpe.py 10000 200 > $OUT/time_$rps.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$rps -- python3 $R/tools/perf_kpe.py 10000 200 > $OUT/f_$rps.log 2>&1
  python3 - <<PY
import csv, glob
v=[float(r["Counter_Value"]) for f in glob.glob("$OUT/f_$rps/*/*counter_collection.csv") for r in csv.DictReader(open(f)) if "kpe_mf" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
t=[l for l in open("$OUT/time_$rps.log") if "kpe_main" in l][-1].strip()
print("rps=$rps  read MB per launch = %.0f (2 x FETCH_SIZE KB)  | %s" % (2*sum(v)/max(1,len(v))*1024/1e6, t))
PY
done

#!/usr/bin/env bash
# one PMC pass of one script: tools/pmc_one.sh <outdir> "<counters>" <script.py> [args]; prints the per-kernel means
set -e
OUT=$1; CNT=$2; shift 2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $OUT -- python3 $R/"$@" > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
python3 - "$OUT" <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "rmx" not in k: continue
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", max(len(v) for v in d.values()))
P

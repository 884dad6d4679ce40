#!/usr/bin/env bash
# Same-box A/B of kpe_bs_ring variants built into ab/bs_<name>.so (tools/ab_kpe_bs.sh build style).
# usage (GPU box): tools/ab_ring.sh "base noroll copy wgs3" [n] [dens]
cd "$(dirname "$0")/.."
names=${1:-base}; n=${2:-10000}; dens=${3:-200}
for round in 1 2; do
  for name in $names; do
    so=$PWD/ab/bs_$name.so
    [ -f "$so" ] || continue
    echo "== $name"
    RMX_SO=$so RMX_BS_RING=1 python3 tools/perf_kpe.py $n $dens | grep -m1 -A2 "launches" | tail -2
  done
  echo "== base PADLDS=12 (three workgroups per CU)"
  RMX_SO=$PWD/ab/bs_base.so RMX_BS_PADLDS=12 RMX_BS_RING=1 python3 tools/perf_kpe.py $n $dens | grep -m1 -A2 "launches" | tail -2
done

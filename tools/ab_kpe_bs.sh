#!/usr/bin/env bash
# Builds ablation variants of the library (ab/*.so, git-ignored) for same-box A/B timings of kpe_bs.
# usage: tools/ab_kpe_bs.sh build   (here, no GPU)   |   tools/ab_kpe_bs.sh run [n] [dens]   (GPU box)
set -e
cd "$(dirname "$0")/.."
variants=(${VARIANTS:-"base:" "d3:-DRMX_BS_DEPTH=3" "nofin:-DRMX_BS_NOFIN" "copy:-DRMX_BS_COPY -DRMX_BS_NOFIN"})
if [ "$1" = build ]; then
  mkdir -p ab
  for v in "${variants[@]}"; do
    name=${v%%:*}; flags=${v#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $flags remap_amd/csrc/api.hip -o ab/bs_$name.so &
  done
  wait
  ls -la ab/
else
  n=${2:-10000}; dens=${3:-200}
  for round in 1 2; do
    for v in "${variants[@]}"; do
      name=${v%%:*}
      for ra in ${RAS:-24 12}; do
        echo "== $name ra=$ra"
        [ -f ab/bs_$name.so ] || continue
        RMX_SO=$PWD/ab/bs_$name.so RMX_KPE_BS=1 RMX_BS_RA=$ra python3 tools/perf_kpe.py $n $dens | grep -m1 -A1 "launches" | tail -1
      done
    done
  done
fi

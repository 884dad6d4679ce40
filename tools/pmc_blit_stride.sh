#!/usr/bin/env bash
# GPU box: FETCH_SIZE of the blit kernels on the bench stream against the order of the binned lists (RMX_BLIT_STRIDE)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for st in ${STRIDES:-0 1 8 16 32 64}; do
  export RMX_BLIT_STRIDE=$st
  for bs in 0 1; do
    export RMX_BLIT_BS=$bs
    echo "== stride $st bit planes $bs"; tools/pmc_one.sh $PWD/gpurun_out/pmcb "FETCH_SIZE" tools/time_blit.py 10000 | grep -i "fgm_blit"
    python3 tools/time_blit.py 10000 2>/dev/null | grep "default:"
  done
done

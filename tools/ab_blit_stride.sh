#!/usr/bin/env bash
# GPU box: the order of the binned lists (RMX_BLIT_STRIDE: 0 = frame order) for both blit kernels
cd "$(dirname "$0")/.."
for st in ${STRIDES:-0 1 16 256 288 320}; do
  export RMX_BLIT_STRIDE=$st
  echo "== stride $st"
  WHAT=${WHAT:-a} python3 tools/ab_blit.py ${N:-10000} ${WHAT:-a}
  RMX_BLIT_BS=0 RMX_SO=$PWD/remap_amd/libremap_amd.so python3 tools/ab_blit.py ${N:-10000} ${WHAT:-a} | sed 's/^libremap_amd.so/fgm_blit (BS=0)/'
  [ -f ab/blit_NOCOMPUTE.so ] && RMX_SO=$PWD/ab/blit_NOCOMPUTE.so python3 tools/ab_blit.py ${N:-10000} ${WHAT:-a}
done

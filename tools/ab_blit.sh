#!/usr/bin/env bash
# GPU box: tools/ab_blit.py with fgm.hip's kernel (default), with the bit-plane kernel (RMX_BLIT_BS=1), and with every
# ab/blit_*.so (BS: the environment's)
cd "$(dirname "$0")/.."
python3 tools/ab_blit.py ${N:-10000} ${WHAT:-ab}
[ -n "$SKIP_BS" ] || RMX_BLIT_BS=1 RMX_SO=$PWD/remap_amd/libremap_amd.so python3 tools/ab_blit.py ${N:-10000} ${WHAT:-ab} | sed 's/^libremap_amd.so/bit planes (BS=1)/'
for so in ab/blit_*.so; do
  [ -f "$so" ] || continue
  RMX_SO=$PWD/$so python3 tools/ab_blit.py ${N:-10000} ${WHAT:-ab}
done

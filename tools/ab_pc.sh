#!/usr/bin/env bash
# GPU box: the default library, then every ab/pc_*.so through tools/ab_pc.py (kpe_pc variants only)
cd "$(dirname "$0")/.."
python3 tools/ab_pc.py ${N:-10000} ${DENS:-200}
for so in ab/pc_*.so; do
  [ -f "$so" ] || continue
  echo "== $so"
  RMX_SO=$PWD/$so python3 tools/ab_pc.py ${N:-10000} ${DENS:-200} pc48:RMX_BS_RING=1 ${EXTRA}
done

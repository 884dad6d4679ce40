#!/usr/bin/env bash
# GPU box: tools/ab_kpm.py with the default library and every ab/kpm_*.so, dense 320x240 and the native 640x480 stream
cd "$(dirname "$0")/.."
for args in "10000 40" "10000 200 640 480 4096" "10000 200"; do
  python3 tools/ab_kpm.py $args
  for so in ab/kpm_*.so; do
    [ -f "$so" ] || continue
    RMX_SO=$PWD/$so python3 tools/ab_kpm.py $args
  done
done

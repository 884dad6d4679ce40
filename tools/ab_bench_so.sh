#!/usr/bin/env bash
# GPU box: the headline of bench.py (three blocks in flight) with the default library and with every ab/*.so,
# two rounds: ms per step and the extraction kernel's time in the timed region.
cd "$(dirname "$0")/.."
for round in 1 2; do
  for so in default ab/*.so; do
    [ "$so" = default ] || [ -f "$so" ] || continue
    if [ "$so" = default ]; then unset RMX_SO; else export RMX_SO=$PWD/$so; fi
    python3 bench.py --headline-only --no-cpu --no-workloads $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('%-22s ms/step %.4f  kernel %s  frac %.3f  value %.2f M' % ('$so', d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['value'] / 1e6))
"
  done
done

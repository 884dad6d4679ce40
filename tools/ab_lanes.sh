#!/usr/bin/env bash
# GPU box: the headline with 1..4 blocks in flight (ms per step, extraction kernel time and fraction in the timed region)
cd "$(dirname "$0")/.."
for round in 1 2; do
  for cfg in ${LANES:-2x1 2x2 3x1 3x2 2x3}; do
    lanes=${cfg%x*}; depth=${cfg#*x}
    python3 bench.py --headline-only --no-cpu --no-workloads --lanes $lanes --depth $depth 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        s = d.get('sustained') or {}
        print('lanes $cfg  ms/step %.4f  kernel %.4f  frac %.3f  value %.2f M   sustained median %.4f kernel %.4f' % (d['ms_per_step'], list(d['kernel_ms'].values())[0], d['roofline']['frac'], d['value'] / 1e6, s.get('ms_per_step_median', 0), s.get('kernel_ms', 0)))
"
  done
done

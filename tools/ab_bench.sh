#!/usr/bin/env bash
# the bench's headline with several builds / environments on one box: tools/ab_bench.sh "name|ENV=VAL ...|so" ...
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  IFS='|' read -r name envs so <<< "$cfg"
  echo "== $name"
  env $envs ${so:+RMX_SO=$PWD/$so} python3 bench.py --steps 20 --warmup 3 --no-cpu --no-workloads --headline-only --lanes ${LANES:-3} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], round(d['value']/1e6,2), round(d['roofline']['achieved']))"
done

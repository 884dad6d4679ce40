#!/usr/bin/env bash
# usage: tools/exp.sh "<cflags>" n dens  -- rebuild with experiment flags on the GPU box and time kpe
set -e
cd $GRAFT_REPO_ROOT
RMX_CFLAGS="$1" python -c "from remap_amd import lib; lib.build(force=True)"
python tools/perf_kpe.py $2 $3 2>&1 | tail -3

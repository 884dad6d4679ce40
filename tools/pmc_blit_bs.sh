#!/usr/bin/env bash
# GPU box: FETCH_SIZE / WRITE_SIZE / instruction counts of the blit kernels on the bench stream (tools/time_blit.py):
# the bit-plane kernel (default) and fgm.hip's (RMX_BLIT_BS=0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  echo "== bit planes: $c"; tools/pmc_one.sh $PWD/gpurun_out/pmcb "$c" tools/time_blit.py 10000 | grep -i blit
  export RMX_BLIT_BS=0
  echo "== fgm_blit: $c"; tools/pmc_one.sh $PWD/gpurun_out/pmcb "$c" tools/time_blit.py 10000 | grep -i blit
  unset RMX_BLIT_BS
done

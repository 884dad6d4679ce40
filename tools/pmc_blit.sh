#!/usr/bin/env bash
# PMC passes for fgm_blit (tools/time_blit.py); usage on the GPU box: bash tools/pmc_blit.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
mkdir -p $OUT
run() { # name, counters
  rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/tools/time_blit.py 10000 > $OUT/$1.log 2>&1 || { tail -5 $OUT/$1.log; }
}
run sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
run sq3 "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH"
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"
run tcc3 "TCC_HIT_sum TCC_MISS_sum"
run tcp1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum"
python3 $R/tools/pmc_to_json.py $OUT fgm_blit $OUT/pmc_fgm_blit.json

#!/usr/bin/env bash
# usage: tools/pmc_any.sh <tag> <script.py> [args...]   (PMC passes for any tool script)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/"${@:3}" > $OUT/$1.log 2>&1 || tail -3 $OUT/$1.log; }
run sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "$@"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "$@"
run sq3 "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_LDS" "$@"
run tcc1 "FETCH_SIZE" "$@"
run tcc2 "WRITE_SIZE" "$@"

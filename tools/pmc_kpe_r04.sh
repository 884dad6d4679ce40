#!/usr/bin/env bash
# PMC passes of the extraction kernel (tools/perf_kpe.py $N: "frames dens [W H]"); separate passes, --kernel-trace only beside them
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1
mkdir -p $OUT
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $OUT/$1 -- python3 $R/tools/perf_kpe.py $N > $OUT/$1.log 2>&1 || { tail -5 $OUT/$1.log; }; }
run sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"

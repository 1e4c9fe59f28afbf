#!/bin/bash
# Counter passes over ONE layer of scripts/perf_conv.py (ONLY=<index>; SCRIPT=perf_convT.py for the transposed
# convolutions): where do the cycles of a kernel go?
#   [SCRIPT=perf_convT.py] bash scripts/pmc_layer.sh <ONLY index> <out dir under gpurun_out>
# One rocprofv3 --pmc run per pass (the hardware holds 8 SQ / 4 TCC / 4 TCP counters at once); csv output.
set -e
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ONLY=$1; OUT=$ROOT/gpurun_out/$2; mkdir -p $OUT
export ONLY
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT -o $name --output-format csv -- python3 $ROOT/scripts/${SCRIPT:-perf_conv.py} > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d $OUT -o trace --output-format csv -- python3 $ROOT/scripts/${SCRIPT:-perf_conv.py} > $OUT/trace.log 2>&1
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_BUSY_avr TA_TA_BUSY_sum
ls $OUT

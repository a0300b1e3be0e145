#!/bin/bash
# PMC passes for one kernel (separate passes; counters only, no trace domains).
# usage: tools/pmc_merge.sh <kernel-regex> <outdir-name> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp
RX="$1"; NAME="$2"; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "$RX" --output-format csv -d $OUT/p$i -o p$i -- python $GRAFT_REPO_ROOT/bench.py --reps 1 --no-cpu-baseline --no-graph "$@" > $OUT/p$i.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head

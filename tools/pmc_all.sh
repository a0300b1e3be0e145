#!/bin/bash
# Counters for EVERY k_* kernel of one bench step (1 stream), one rocprofv3 --pmc pass per counter group (counters only).
# usage: tools/pmc_all.sh <outdir-name>
set -u
NAME=$1
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
rm -rf $OUT && mkdir -p $OUT
ARGS="--no-cpu-baseline --no-h2d --no-graph --streams 1 --steps 1 --warmup 1"
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-include-regex "k_" --output-format csv -d $OUT/p$i -o p -- python $GRAFT_REPO_ROOT/bench.py --reps 1 $ARGS > $OUT/p$i.log 2>&1
done
rm -rf /tmp/pmc_all_kt
rocprofv3 --kernel-trace -d /tmp/pmc_all_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --reps 1 $ARGS > /dev/null 2>&1
cp $(find /tmp/pmc_all_kt -name "*results.db" | head -1) $OUT/kt.db
python $GRAFT_REPO_ROOT/tools/pmc_all_report.py $OUT 2 > $OUT/table.md
rm -f $OUT/kt.db
find $OUT -name "*agent_info*" -delete

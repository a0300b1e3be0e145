#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE; counters only) + one kernel trace of the same bench command.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/hbm
rm -rf $OUT && mkdir -p $OUT
ARGS="--no-cpu-baseline --no-c5 --streams 1 --steps 1 --warmup 1"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-include-regex "k_" --output-format csv -d $OUT/$C -o p -- python $GRAFT_REPO_ROOT/bench.py --reps 1 $ARGS > $OUT/$C.log 2>&1
done
rocprofv3 --kernel-trace -d /tmp/hbm_kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --reps 1 $ARGS > /dev/null 2>&1
cp /tmp/hbm_kt/kt_results.db $OUT/kt.db
python $GRAFT_REPO_ROOT/tools/hbm_table.py $OUT 2 > $OUT/table.md
rm -f $OUT/kt.db
find $OUT -name "*agent_info*" -delete
du -sh $OUT

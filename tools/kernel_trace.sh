#!/bin/bash
# Kernel traces of bench.py (one per argument set) -> gpurun_out/<name>/{default,1stream}.md
# usage: tools/kernel_trace.sh <outdir-name> [steps]
set -u
NAME=$1; STEPS=${2:-5}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$NAME
mkdir -p $OUT
for V in default 1stream; do
  EXTRA=""; [ $V = 1stream ] && EXTRA="--streams 1"
  rm -rf /tmp/kt_$V
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$V -o kt -- python $GRAFT_REPO_ROOT/bench.py --reps 1 --no-cpu-baseline --no-h2d --no-graph --steps $STEPS --warmup 2 $EXTRA > $OUT/$V.log 2>&1
  DB=$(find /tmp/kt_$V -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB $((STEPS + 2)) > $OUT/$V.md
done
tail -1 $OUT/default.log | cut -c1-400

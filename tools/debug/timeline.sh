#!/bin/bash
# kernel timeline of graph-replayed steps: busy / idle / kernels in flight (tools/debug/timeline.py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --reps 1 --no-cpu-baseline --no-h2d --steps 6 --warmup 3 "$@" > /tmp/tl.log 2>&1
python $GRAFT_REPO_ROOT/tools/debug/timeline.py $(find /tmp/tl -name "*results.db" | head -1) 3

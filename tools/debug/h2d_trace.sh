#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for m in "$@"; do
  rm -rf /tmp/h2d_$m
  rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/h2d_$m -o t -- python $GRAFT_REPO_ROOT/tools/debug/h2d_trace.py run $m > /tmp/h2d_$m.log 2>&1
  grep "^step" /tmp/h2d_$m.log || tail -15 /tmp/h2d_$m.log
  python $GRAFT_REPO_ROOT/tools/debug/h2d_trace.py report $(find /tmp/h2d_$m -name "*results.db" | head -1)
done

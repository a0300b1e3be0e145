#!/bin/bash
# early vs late fork of the side streams (BurstPipeline._on_streams, round 5): single-GPU step and per-rank compute at G = 8
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in late early; do
    if [ $v = late ]; then export HHSR_LATE_FORK=1; else unset HHSR_LATE_FORK; fi
    python bench.py --no-cpu-baseline --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v fork: step', d['ms_per_step'], 'eager', d['ms_per_step_eager'])"
  done
done
for v in late early; do
  if [ $v = late ]; then export HHSR_LATE_FORK=1 HHSR_ROWS_SPLIT_REF=1; else unset HHSR_LATE_FORK HHSR_ROWS_SPLIT_REF; fi
  python tools/debug/emulate_ranks.py --worlds 4,8 --steps 10 --strategies rows 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', 'G =', d['world'], 'max rank ms', d['max_rank_ms'], 'mean', d['mean_rank_ms'], [ (r['ms'], r['ms_A_alone'], r['ms_B_alone']) for r in d['per_rank'][:4]])"
done

#!/bin/bash
# Round 5's overlap changes of the per-rank step, A/B on one MI355X (environment switches read at capture):
#   r4    HHSR_LATE_FORK=1 HHSR_ROWS_SPLIT_REF=1   round 4: side streams fork behind the reference precompute, reference
#                                                  alignment state as its own graph
#   fork  (default)                                fork at the start of the precompute, reference state inside the first step-A graph
# (profiles/r05_rows_overlap_ab.txt also has "pre": step B's raw pass — it needs no flow — captured as its own graph on the
#  step-B stream next to step A; no gain: 1.540 vs 1.542 ms at G = 8, 2.529 vs 2.515 at G = 4; removed)
# single-GPU step (bench.py, graph replay / eager) for r4 and default; per-rank compute at G = 4, 8 (tools/debug/emulate_ranks.py)
cd $GRAFT_REPO_ROOT
setv() { unset HHSR_LATE_FORK HHSR_ROWS_SPLIT_REF
  case $1 in r4) export HHSR_LATE_FORK=1 HHSR_ROWS_SPLIT_REF=1;; esac; }
if [ "${1:-all}" != ranks ]; then
for i in 1 2 3; do
  for v in r4 fork; do
    setv $v
    python bench.py --no-cpu-baseline --no-h2d --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v: step', d['ms_per_step'], 'eager', d['ms_per_step_eager'])"
  done
done
fi
for v in r4 fork; do
  setv $v
  python tools/debug/emulate_ranks.py --worlds ${WORLDS:-4,8} --steps 10 --strategies rows 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', 'G =', d['world'], 'max rank ms', d['max_rank_ms'], 'mean', d['mean_rank_ms'], [ (r['ms'], r['ms_A_alone'], r['ms_B_alone']) for r in d['per_rank'][:4]])"
done

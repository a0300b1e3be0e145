cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 20 2>&1 | grep "^{" > $O/emulate_ranks_c3.jsonl
python -c "
import json
for l in open('$O/emulate_ranks_c3.jsonl'):
    d=json.loads(l); print(d['world'], d['strategy'], d['max_rank_ms'], [r['ms'] for r in d['per_rank']])
"

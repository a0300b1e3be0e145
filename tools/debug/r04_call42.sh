cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 10 2>&1 | grep "^{" > $O/emulate_ranks_c3.jsonl
python tools/debug/emulate_ranks.py --worlds 2,4,8 --steps 10 --strategies rows --stage-frames 4 2>&1 | grep "^{" > $O/emulate_ranks_c3_staged.jsonl
python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 3 --height 6000 --width 8000 --scale 3 --strategies rows 2>&1 | grep "^{" > $O/emulate_ranks_c5.jsonl
python -c "
import json
for f in ('c3','c3_staged','c5'):
    for l in open('$O/emulate_ranks_%s.jsonl'%f):
        d=json.loads(l); print(f, d['world'], d['strategy'], d['max_rank_ms'], d.get('modelled_reduce_scatter_ms'))
"

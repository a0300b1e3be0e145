cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 1800 python -m pytest tests/test_hip_parity.py -q -m gpu -k "graph or host_burst or runner or two_ranks or sharded or rccl or rows_plan or bench or e2e_golden_128" -p no:cacheprovider 2>&1 | tail -3
python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 10 2>&1 | grep "^{" > $O/emulate_ranks_c3.jsonl
python tools/debug/emulate_ranks.py --worlds 2,4,8 --steps 10 --strategies rows --stage-frames 4 2>&1 | grep "^{" > $O/emulate_ranks_c3_staged.jsonl
python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 3 --height 6000 --width 8000 --scale 3 --strategies rows 2>&1 | grep "^{" > $O/emulate_ranks_c5.jsonl
python -c "
import json
for f in ('c3','c3_staged','c5'):
    for l in open('$O/emulate_ranks_%s.jsonl'%f):
        d=json.loads(l); print(f, d['world'], d['strategy'], d['max_rank_ms'], d.get('modelled_reduce_scatter_ms'))
"
python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | cut -c1-300

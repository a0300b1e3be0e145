cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "two_ranks or sharded or rccl or rows_plan" -p no:cacheprovider 2>&1 | tail -3
python tools/debug/emulate_ranks.py --worlds 1,8 --steps 10 --strategies rows 2>&1 | grep "^{" > gpurun_out/r04/emulate_ranks_c3_b.jsonl
python tools/debug/emulate_ranks.py --worlds 1,8 --steps 3 --height 6000 --width 8000 --scale 3 --strategies rows 2>&1 | grep "^{" > gpurun_out/r04/emulate_ranks_c5_b.jsonl
python -c "
import json
for f in ('c3','c5'):
    for l in open('gpurun_out/r04/emulate_ranks_%s_b.jsonl'%f):
        d=json.loads(l); print(f, d['world'], d['max_rank_ms'], [ (r['rows'][1]-r['rows'][0], r['ms']) for r in d['per_rank']])
"

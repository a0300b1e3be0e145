cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests/test_hip_parity.py -q -m gpu -k "sharded or rccl or two_ranks or corners or robustness or rob_" -p no:cacheprovider) > gpurun_out/r04/t19.log 2>&1
tail -4 gpurun_out/r04/t19.log
python tools/debug/emulate_ranks.py --worlds 8 --steps 10 --strategies rows 2>&1 | grep "^{" > gpurun_out/r04/emu_c3_final.jsonl
python tools/debug/emulate_ranks.py --worlds 8 --steps 3 --strategies rows --height 6000 --width 8000 --scale 3 2>&1 | grep "^{" > gpurun_out/r04/emu_c5_final.jsonl
python - <<'PY'
import json
for f in ("emu_c3_final","emu_c5_final"):
    for l in open(f"gpurun_out/r04/{f}.jsonl"):
        d=json.loads(l); print(f, d["world"], d["max_rank_ms"], d["mean_rank_ms"], [(r["rows"][1]-r["rows"][0], r["ms"]) for r in d["per_rank"]])
PY

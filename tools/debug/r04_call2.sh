cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(time timeout 2400 python -m pytest tests/test_hip_parity.py -q -m gpu -k "rccl or corners or thread_safe or sharded_hip_engine or seam or batched" -p no:cacheprovider) > gpurun_out/r04/t2.log 2>&1
tail -5 gpurun_out/r04/t2.log
(time timeout 900 python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 10 --strategies rows) > gpurun_out/r04/emu_c3_b.log 2>&1
(time timeout 900 python tools/debug/emulate_ranks.py --worlds 2,8 --steps 10 --strategies rows --stage-frames 100) > gpurun_out/r04/emu_c3_onestage.log 2>&1
(time timeout 1500 python tools/debug/emulate_ranks.py --worlds 1,2,4,8 --steps 3 --height 6000 --width 8000 --scale 3) > gpurun_out/r04/emu_c5.log 2>&1
python - <<'PY'
import json
for f in ("emu_c3_b", "emu_c3_onestage", "emu_c5"):
    print(f)
    for l in open(f"gpurun_out/r04/{f}.log"):
        if l.startswith("{"):
            d = json.loads(l)
            print(" ", d["world"], d["strategy"], d["max_rank_ms"], [(r.get("ms"), r.get("ms_A_alone"), r.get("ms_B_alone")) for r in d["per_rank"]][:8])
        elif "Error" in l or "error" in l:
            print(l.strip()[:300])
PY

"""Per-rank compute of the multi-GPU strategies, measured on ONE GPU: for G in --worlds, every rank j of G runs ITS part
of the step on this GPU alone, replayed from the same HIP graphs a real G-GPU job replays (distributed.RowsPlan for
"rows": step A of the rank's frames on one stream, step B of its slab on another; the all-gather is replaced by a device
copy of the flows a single-GPU alignment computed) — so max_j T_j(G) is the step time of a G-GPU job minus its RCCL time
(376 kB per frame and stage for "rows"; the packed accumulators for "reduce").  Prints one JSON line per (G, strategy).

    python tools/debug/emulate_ranks.py [--height 3000 --width 4000 --frames 20 --scale 2] [--worlds 1,2,4,8] [--steps 10]
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import handheld_super_resolution as hsr  # noqa: E402
from handheld_super_resolution import distributed as hdist, synthetic as synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--width", type=int, default=4000)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--scale", type=float, default=2)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--stage-frames", type=int, default=0, help="frames per stage (0 = one stage: the library default)")
    ap.add_argument("--align-cost", type=float, default=None)
    ap.add_argument("--strategies", default="rows,reduce")
    a = ap.parse_args()
    H, W, NF = a.height, a.width, a.frames
    scale = int(a.scale) if float(a.scale).is_integer() else a.scale
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ref, comp, _ = synth.make_burst_torch(H, W, NF, dev, seed=1234)
    n = NF - 1

    def config():
        cfg = hsr.default_config()
        cfg.verbose = 0
        cfg.scale = scale
        cfg.hip = {"stage_frames": a.stage_frames}
        sf = a.stage_frames if a.stage_frames > 0 else max(n, 1)
        if a.align_cost is not None:
            cfg.hip["align_cost"] = a.align_cost
        hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                           [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
        return cfg

    # the flows every rank would receive from the all-gathers
    cfg = config()
    e0 = hdist.HipEngine(cfg).init_ref(ref)
    flows = e0.align_frames([comp[i] for i in range(n)])
    bound = float(flows[..., 1].abs().max())
    sH, sW = round(scale * H), round(scale * W)
    del e0

    def timed(fn, steps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    # Plans and engines stay alive until their (world, strategy) row is printed and are then collected OUTSIDE any capture: a CUDAGraph finalised (hipGraphDestroy) by a garbage collection
    # that happens to run while the NEXT plan's streams are capturing aborts the process ("operation not permitted when
    # stream is capturing" from a destructor).  The product keeps its plans in the engine for the same reason.
    keep = []
    for G in [int(g) for g in a.worlds.split(",")]:
        for strategy in a.strategies.split(","):
            per_rank, detail = [], []
            for j in range(G):
                cfg = config()
                eng = hdist.HipEngine(cfg)
                if strategy == "rows":
                    cost = float(cfg.hip.get("align_cost", hdist.align_cost(scale)))
                    stages = hdist.stage_plan(n, G, a.stage_frames if a.stage_frames > 0 else max(n, 1))
                    bounds = hdist.slab_bounds(sH, G, n, cost, hdist.slab_align(scale))
                    r0, r1 = bounds[j], bounds[j + 1]
                    if r1 <= r0:
                        per_rank.append(0.0)
                        continue
                    plan = hdist.RowsPlan(eng, ref, comp, stages, j, G, (r0, r1), 1.25 * bound + 2.0, check=True, key=None)

                    pre = []  # what each stage's all-gather delivers
                    for s, st in enumerate(stages):
                        t = torch.zeros_like(plan.gath[s])
                        for i in hdist.stage_frames(st, n, G):
                            t[i % G, i // G - st[0]].copy_(flows[i])
                        pre.append(t)

                    def step():
                        ctx = plan.open()
                        for s, st in enumerate(stages):
                            ctx.align(s)
                            with torch.cuda.stream(plan.s_b):  # stand-in for the all-gather: the stage's flows appear
                                plan.gath[s].copy_(pre[s], non_blocking=True)
                            ctx.front(s, None)
                        out = ctx.finish()
                        assert not out[3]
                        return out

                    # step A alone / step B alone (the two streams' work, serialised) for the model's terms
                    def only_a():
                        plan.s_a.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(plan.s_a):
                            if plan.g_ref_a is not None:  # (else: inside the first stage's graph)
                                plan.g_ref_a.replay()
                            for g in plan.g_a:
                                if g is not None:
                                    g.replay()
                        torch.cuda.current_stream().wait_stream(plan.s_a)

                    def only_b():
                        plan.s_b.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(plan.s_b):
                            plan.g_ref_b.replay()
                            for g in plan.g_b:
                                g.replay()
                            plan.g_fin.replay()
                        torch.cuda.current_stream().wait_stream(plan.s_b)

                    step()
                    t = timed(step, a.steps)
                    ta, tb = timed(only_a, a.steps), timed(only_b, a.steps)
                    detail.append({"rank": j, "rows": [r0, r1], "frames_aligned": len(range(j, n, G)), "ms": round(t, 3),
                                   "ms_A_alone": round(ta, 3), "ms_B_alone": round(tb, 3)})
                    keep.append(plan)  # (see `keep`)
                else:
                    rows = hdist.slab_rows(sH, G)
                    bounds = hdist.slab_bounds(sH, G)
                    r0, r1 = bounds[j], bounds[j + 1]
                    mine = [comp[i] for i in range(j, n, G)]
                    red = torch.zeros((2, rows, sW, 3), dtype=torch.float32, device=dev)

                    def step():
                        acc, acc_r, ref_dev, ref_covs = eng.partial(ref, mine, bounds, rows)
                        red.copy_(acc[j])  # stand-in for the reduce-scatter (its RCCL time is modelled, not measured)
                        return eng.finish_rows(red, r0, r1, ref_dev, ref_covs) if r1 > r0 else None

                    for _ in range(3):
                        step()
                    t = timed(step, a.steps)
                    detail.append({"rank": j, "rows": [r0, r1], "frames": len(mine), "ms": round(t, 3)})
                per_rank.append(t)
                keep.append(eng)
                torch.cuda.empty_cache()
            rec = {"workload": f"{H}x{W}x{NF} x{scale}", "world": G, "strategy": strategy, "stage_frames": a.stage_frames,
                   "max_rank_ms": round(max(per_rank), 3), "mean_rank_ms": round(float(np.mean(per_rank)), 3),
                   "per_rank": detail}
            if strategy == "reduce":
                gb = 2 * sH * sW * 3 * 4 / 1e9
                rec["accumulator_GB_per_rank"] = round(gb, 3)
                rec["modelled_reduce_scatter_ms"] = round(gb * (G - 1) / G / 100.0 * 1e3, 2) if G > 1 else 0.0
            print(json.dumps(rec), flush=True)
            plan = eng = None
            keep.clear()
            torch.cuda.synchronize()
            gc.collect()


if __name__ == "__main__":
    main()

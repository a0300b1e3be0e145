#!/usr/bin/env python
"""Benchmark of the burst super-resolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (the reference's main(): grey -> pyramid block matching -> ICA ->
robustness -> kernel estimation -> merge -> normalise) over one synthetic RAW burst whose frames are already resident
in HBM.  Default workload = the configuration BASELINE.json's metric is quoted on: 3000x4000 (12 MP), 20 frames, x2 ->
48 MP output, full align + ICA + robustness + merge (it fits one MI355X: < 3 GB resident).

N > 1: one process per GPU over RCCL (backend "nccl").  The driver launches the ranks with torch.distributed.run; run
by hand (`python bench.py --gpus 4`) the script re-executes itself under torch.distributed.run.  Every rank count —
N = 1 included — goes through handheld_super_resolution.distributed.main_sharded.  --strategy rows (default): alignment
frame-parallel, ONE all-gather of the flow fields, kernels / robustness / merge row-parallel; --strategy reduce (the north
star's): frames one per rank, ONE reduce-scatter of the float32 accumulators.  The output stays sharded by rows
(`--gather` adds the gather to rank 0).  Total work is fixed, so scaling is "strong".

Rank 0 prints ONE JSON line: metric "output Mpix/s" (scale^2 * H * W / time per burst), plus
  value_incl_h2d  the same with the frames starting in HOST memory (the reference's timer spans its uploads:
                  super_resolution.py:103-195): page-locked float32 arrays (with pcie_floor_ms / pcie_floor_frac),
                  h2d.*_u16: uint16 sensor counts normalised on the device (half the PCIe bytes), h2d.value_numpy_pageable:
                  plain NumPy float32 arrays through main() itself — eager uploads into static staging, per-chunk HIP
                  graphs, chained fused merge (handheld_super_resolution/graph.py: HostBurstRunner);
  roofline        dominant kernel (hhsr_merge_burst): VALU issue and algorithmic bytes per launch / launch duration
                  measured with HIP events on the launch stream;
  cpu_baseline    the NumPy oracle (a golden-pinned port of the reference's algorithm; the reference itself has no CPU
                  path) on every core the container may use (logical CPUs capped by the cgroup CPU quota; one worker
                  process per comp frame, several crops side by side when the cores allow) on a bounded crop of the burst;
  parity          max-abs difference of the GPU result to the oracle on that crop, with the residual attributed:
                  tiles whose block-matching decision differs (float32 near-ties), and the difference that remains
                  when the oracle's flow fields are injected into the GPU path.
"""
import argparse
import hashlib
import importlib.util
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.join(ROOT, "handheld-multi-frame-super-resolution_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

VALU_PEAK_TLANEOPS = 78.6  # 256 CUs x 4 SIMD-32 x 2.4 GHz: one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MERGE_SRC = {2.0: ("hhsr_merge.h", "hhsr_merge_x2.hip"), 3.0: ("hhsr_merge.h", "hhsr_merge_xs.hip")}  # kernel sources (csrc/)
LEG_TIMEOUT_S = 300        # N > 1: an optional leg whose collectives hang must not lose the line (watchdog)


def merge_burst_bytes(n_comp, P, S):
    """Algorithmic HBM bytes of ONE whole-image hhsr_merge_burst launch (fp32): per comp frame raw + covariances +
    robustness = 12 P (flow is negligible); reference frame raw + covariances = 8 P; output 12 S P."""
    return float(n_comp * 12 * P + 8 * P + 12 * S * P)


def step_bytes(n_comp, P, S):
    """Algorithmic HBM bytes of one WHOLE step of the fused pipeline (fp32, Bayer; DESIGN.md §4 kernel table), per comp
    frame: grey FFT 16 P (raw in, kept half spectrum out / in / out / in, grey out); pyramid 4 P + P + P/16 + P/256 read,
    P/4 + P/64 + P/1024 written ~ 5.4 P; alignment reference + moving level read once per level, flow negligible:
    8 P (1 + 1/4 + 1/64 + 1/1024) ~ 10.1 P; raw pass 4 P in, 3 P means + 4 P covariances out; robustness 3 P means + 4 P
    R out + 20 P of reference planes per group of 4 frames = 12 P; merge 12 P.  Per burst: the merge's reference frame
    and output (8 P + 12 S P), the reference frame's own precompute ~ 60 P.  (SURVEY.md §8d's 68.5 GB model priced the
    reference's per-frame read-modify-write of the accumulators, which the fused merge does not do.)"""
    per_frame = (16 + 5.4 + 10.1 + 11 + 12 + 12) * P
    return float(n_comp * per_frame + (8 + 12 * S) * P + 60 * P)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=3000)
    ap.add_argument("--width", type=int, default=4000)
    ap.add_argument("--frames", type=int, default=20, help="burst length including the reference frame")
    ap.add_argument("--scale", type=float, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline and the parity leg")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-resident (H2D-inclusive) leg")
    ap.add_argument("--cpu-crop", type=int, default=1024, help="edge of the crop the CPU baseline / parity leg runs on")
    ap.add_argument("--streams", type=int, default=None, help="HIP streams of the frame pipeline (default: config, 2; 1 for frames of 40 MP and more)")
    ap.add_argument("--chunk", type=int, default=None, help="frames per front-end chunk = per batched launch (default 4)")
    ap.add_argument("--gather", action="store_true", help="N > 1: gather the finished row slabs to rank 0")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for plumbing tests)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python instead of replaying the "
                                                            "step from a HIP graph")
    ap.add_argument("--engine", default=None, help="FILE.py:CLASS replacing distributed.HipEngine (plumbing tests "
                                                   "without a GPU; implies host tensors; only with --backend gloo)")
    ap.add_argument("--strategy", default="rows", choices=["rows", "reduce"],
                    help="N > 1: rows = frame-parallel alignment, all-gather of flows, row-parallel merge (default); "
                         "reduce = frames one per rank, ONE reduce-scatter of the float32 accumulators (north star)")
    ap.add_argument("--max-flow", type=float, default=None,
                    help="N > 1, rows: bound on |flow_y| in pixels for the sub-image halo (no host read in the step); "
                         "default: measured from the gathered flows (one scalar read per burst)")
    ap.add_argument("--cpu-cores", type=int, default=None, help="worker processes of the CPU baseline (default: all cores)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed block of --steps steps: ms_per_step is "
                                                        "the MEDIAN block, ms_per_step_min / _max the spread")
    ap.add_argument("--no-c5", action="store_true", help="skip the optional 48 MP x 20 x3 (C5 geometry) leg")
    ap.add_argument("--denoiser", action="store_true", help="the main workload with the accumulated-robustness merge denoiser on "
                    "(kernel traces of that path; the default run times it as the ms_per_step_denoiser leg)")
    ap.add_argument("--weight-fp64", action="store_true", help="the main workload with config.hip.weight_fp64 (the reference's "
                                                               "float64 weight chain on every pixel; kernel traces)")
    args = ap.parse_args()
    if args.engine is not None and args.backend != "gloo":
        ap.error("--engine (a foreign per-rank engine: launch-plumbing tests) is only accepted with --backend gloo")
    return args


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` by hand: become `python -m torch.distributed.run --nproc-per-node N bench.py ...`."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def load_engine(spec):
    path, cls = spec.rsplit(":", 1)
    mod_spec = importlib.util.spec_from_file_location("bench_engine", path)
    mod = importlib.util.module_from_spec(mod_spec)
    sys.path.insert(0, os.path.dirname(os.path.abspath(path)))
    mod_spec.loader.exec_module(mod)
    return getattr(mod, cls)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)  # does not return

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    on_gpu = args.engine is None
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py: no HIP device visible (the hot path has no CPU fallback)")
        # (HHSR_BENCH_SHARE_GPU=1: every rank on device 0 — lets a 1-GPU box exercise the multi-rank path)
        share = os.environ.get("HHSR_BENCH_SHARE_GPU") == "1"
        torch.cuda.set_device(local_rank if (world > 1 and not share) else 0)
        dev = torch.device("cuda", torch.cuda.current_device())
    else:
        dev = torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus

    import handheld_super_resolution as hsr
    from handheld_super_resolution import synthetic as synth, distributed as hdist, merge as hmerge

    H, W, NF = args.height, args.width, args.frames
    scale = int(args.scale) if float(args.scale).is_integer() else args.scale
    # identical burst on every rank (deterministic generator); stays resident in HBM
    if on_gpu:
        ref, comp, shifts = synth.make_burst_torch(H, W, NF, dev, seed=1234)
    else:
        r_, c_, shifts = synth.make_burst(H, W, NF, seed=1234)
        ref, comp = torch.from_numpy(r_), torch.from_numpy(c_)
    cfg = hsr.default_config()
    cfg.verbose = 0
    cfg.scale = scale
    cfg.hip = {"graph": not args.no_graph}
    if args.weight_fp64:
        cfg.hip["weight_fp64"] = True
    if args.denoiser:
        cfg.accumulated_robustness_denoiser.enabled = True
        cfg.accumulated_robustness_denoiser.merge.enabled = True
    if args.streams is not None:
        cfg.hip["streams"] = args.streams
    if args.chunk is not None:
        cfg.hip["chunk"] = args.chunk
    hsr.prepare_config(cfg, np.full((H, W), float(ref.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                       [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
    engine_cls = hdist.HipEngine if on_gpu else load_engine(args.engine)

    # time the dominant kernel with HIP events on the stream it is launched on (torch's current stream)
    ev = []
    orig_call = hmerge._lib.call

    def timed_call(name, *a):
        if name == "hhsr_merge_burst" and timed_call.on and on_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig_call(name, *a)
            e1.record()
            timed_call.sink.append((e0, e1, a[4]))
        else:
            orig_call(name, *a)

    timed_call.on = False
    timed_call.sink = ev
    timed_call.steps = 0
    hmerge._lib.call = timed_call

    # shader clock DURING a timed block: a one-wave probe (hhsr_clock_probe) sleeps on a side stream next to the block's
    # kernels for ~60 % of its expected duration and reports shader cycles per 100 MHz tick
    probe_stream = torch.cuda.Stream() if on_gpu else None

    def probe_start(expect_ms):
        if not on_gpu or expect_ms <= 0:
            return None
        buf = torch.zeros(4, dtype=torch.int64, device=dev)
        with torch.cuda.stream(probe_stream):
            hmerge._lib.call("hhsr_clock_probe", hmerge._lib.ptr(buf), int(0.6 * expect_ms * 1e5), hmerge._lib.stream())
        return buf

    def probe_mhz(buf):
        if buf is None:
            return None
        probe_stream.synchronize()
        c0, r0, c1, r1 = (int(v) for v in buf.cpu().tolist())
        return (c1 - c0) / (r1 - r0) * 100.0 if r1 > r0 else None

    def timed_reps(fn, steps, reps, warmup):
        """`reps` timed blocks of `steps` steps (each bracketed like the contract says): per-block ms per step.  The shader
        clock is measured in ONE MORE block behind them, with the probe kernel sleeping on a side stream next to the block's
        kernels: a second active queue costs the step ~2 % (first version: probe next to blocks 2 - 5, which were all 2 %
        slower than block 1), so that block's time is reported apart (clock_block_ms) and never enters the median."""
        blocks, clocks = [], []
        if reps > 1:  # one untimed block first (the chip's clock / power state settles ~150 ms into a run)
            timed(fn, steps, warmup)
            warmup = 0
        for rep in range(max(1, reps)):
            blocks.append(timed(fn, steps, warmup if rep == 0 else 0))
        if reps > 1:
            try:
                pb = probe_start(blocks[-1] * steps)
                t_probe = timed(fn, steps, 0)
                mhz = probe_mhz(pb)
                if mhz:
                    clocks.append(mhz)
                timed_reps.clock_block_ms = t_probe
            except Exception as e:  # noqa: BLE001
                errors.setdefault("clock_probe", f"{type(e).__name__}: {e}")
        return blocks, clocks

    timed_reps.clock_block_ms = None

    def spread(blocks):
        b = sorted(blocks)
        return {"median": b[len(b) // 2] if len(b) % 2 else 0.5 * (b[len(b) // 2 - 1] + b[len(b) // 2]), "min": b[0], "max": b[-1]}

    def pmc_lookup(scale_, H_, W_, NF_, launch_ms):
        """HBM bytes / VALU instructions per launch of the merge kernel from the committed rocprofv3 --pmc passes of this
        exact kernel source (newest profiles/rNN_pmc_merge*.json first); (None, None, why) when the record is stale."""
        name = "pmc_merge_x3.json" if float(scale_) == 3.0 else "pmc_merge.json"
        h = hashlib.sha256()
        for src in MERGE_SRC.get(float(scale_), ("hhsr_merge.h", "hhsr_merge.hip")):
            h.update(open(os.path.join(PKG_ROOT, "csrc", src), "rb").read())
        sha = h.hexdigest()[:16]
        note = "no PMC record in profiles/"
        for fname in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_" + name)), reverse=True):
            try:
                with open(os.path.join(ROOT, "profiles", fname)) as f:
                    pm = json.load(f)
            except Exception as e:  # noqa: BLE001
                note = f"profiles/{fname}: {e}"
                continue
            if pm.get("workload") != f"{H_}x{W_}x{NF_} x{scale_}":
                continue
            if pm.get("source_sha16") != sha:
                note = (f"profiles/{fname} was collected for kernel source {pm.get('source_sha16')}, the source is now "
                        f"{sha}: counters not reported (re-run tools/pmc_merge.sh)")
                continue
            insts = pm["valu_wave_insts_per_launch"]
            lane_ops = insts * 64 / (launch_ms * 1e-3)
            return pm["traffic_bytes_per_launch"], {
                "wave_insts_per_launch": insts, "achieved_Tlaneops": round(lane_ops / 1e12, 2),
                "peak_Tlaneops": VALU_PEAK_TLANEOPS, "frac": round(lane_ops / 1e12 / VALU_PEAK_TLANEOPS, 3),
                "source": f"profiles/{fname}"}, None
        return None, None, note

    # one engine for all steps: with device-resident frames it captures the step in a HIP graph on its second call and
    # replays it afterwards (handheld_super_resolution/graph.py) — one launch per burst instead of ~280
    engine = engine_cls(cfg)

    def step(r=ref, c=comp):
        timed_call.steps += int(timed_call.on)
        return hdist.main_sharded(r, c, cfg, engine=engine, gather=args.gather, strategy=args.strategy,
                                  max_flow=args.max_flow)[0]

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            out = fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        barrier()
        dt = time.perf_counter() - t0
        del out
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / steps * 1e3

    errors = {}
    t_region = time.perf_counter()
    for _ in range(args.warmup):
        step()
    barrier()
    timed_call.on = True
    # EXACTLY --steps steps per timed block, --reps blocks back to back: the headline is the median block
    ms_blocks, sclk = timed_reps(step, args.steps, args.reps, 0)
    timed_call.on = False
    timed_region_s = time.perf_counter() - t_region
    sp = spread(ms_blocks)
    ms_per_step = sp["median"]
    out_pix = round(scale * H) * round(scale * W)
    value = out_pix / (ms_per_step * 1e-3) / 1e6
    P, S = H * W, float(scale) ** 2
    headline = NF == 20 and H * W == 12_000_000 and scale == 2
    # ---- the measurement exists from here on: the line is built NOW and printed exactly once — at the end of the optional
    # legs below (each guarded: a failing leg leaves its error string in line["errors"]), or by the exit hook if anything
    # else ends the process first.  N > 1 runs none of the host-resident / CPU legs (they are N = 1 figures).
    line = {
        "metric": "output Mpix/s for 12MP x 20-frame x2 SR burst; max-abs diff vs reference" if headline else
                  f"output Mpix/s for {H * W / 1e6:.0f}MP x {NF}-frame x{scale} SR burst; max-abs diff vs reference",
        "value": round(value, 2), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": round(value / 12.0, 2) if headline else None,
        "dtype": "f32", "data": "synthetic",
        "reps": len(ms_blocks), "ms_per_step_min": round(sp["min"], 3), "ms_per_step_max": round(sp["max"], 3),
        "ms_per_step_blocks": [round(b, 3) for b in ms_blocks],
        "spread_pct": round(100.0 * (sp["max"] - sp["min"]) / sp["median"], 2),
        "sclk_mhz": round(spread(sclk)["median"], 1) if sclk else None,
        "sclk_measured_in": ("one extra block of --steps steps behind the timed ones, the probe kernel on a side stream next to it "
                             f"(that block: {timed_reps.clock_block_ms:.3f} ms per step; not part of the median)"
                             if timed_reps.clock_block_ms else None),
        "timed_region_s": round(timed_region_s, 2),
        "settle_steps": args.steps if len(ms_blocks) > 1 else 0,  # untimed block between the warm-up and the timed blocks
    }
    printed = []

    def emit():
        if rank == 0 and not printed:
            printed.append(1)
            if errors:
                line["errors"] = errors
            print(json.dumps(line), flush=True)

    import atexit

    atexit.register(emit)
    graphed = bool(on_gpu and (any(getattr(getattr(engine, r, None), "graphs", None) for r in ("_runner", "_runner_p"))
                               or getattr(engine, "_plans", None)))
    if world > 1:  # the eager section below runs collectives: every rank takes it when any rank replayed a graph
        g_any = torch.tensor([int(graphed)], dtype=torch.int32, device=dev)
        dist.all_reduce(g_any, op=dist.ReduceOp.MAX)
        graphed = bool(int(g_any.item()))
    ms_eager, ev_steps = None, timed_call.steps  # (not graphed: every step of the timed blocks launched the kernel itself)
    if graphed:
        # the timed steps were graph replays: no Python launch to bracket with events.  The dominant kernel's launch
        # duration comes from a few eager steps of the same workload right after (same kernel, same inputs)
        import copy

        try:
            cfg_e = copy.deepcopy(cfg)
            cfg_e.hip = dict(cfg_e.hip, graph=False)
            eng_e = engine_cls(cfg_e)
            ev_steps = max(3, min(10, args.steps))
            fn_e = lambda: hdist.main_sharded(ref, comp, cfg_e, engine=eng_e, gather=args.gather, strategy=args.strategy,  # noqa: E731
                                              max_flow=args.max_flow)[0]
            for _ in range(2):  # (the second call still grows the caching allocator at 48 MP frames: 86 vs 68 ms per step)
                fn_e()
            barrier()
            timed_call.on = True
            ms_eager = timed(fn_e, ev_steps, 0)
            del eng_e
        except Exception as e:  # noqa: BLE001
            errors["eager_leg"] = f"{type(e).__name__}: {e}"
            ev.clear()
        timed_call.on = False

    # ---- N > 1: BOTH strategies in one record (the driver passes --gpus N only: the north star's `reduce` curve would not
    # exist otherwise), the per-rank compute without collectives measured live, and the committed single-GPU emulation of
    # this rank count next to it — so a SCALE value can be checked against (compute per rank) + (RCCL time).  Optional leg:
    # guarded on every rank, and a watchdog prints the line and ends the process if a collective of this leg hangs.
    strategies, emulated = None, None
    if world > 1:
        import threading

        def give_up():
            errors["strategies_leg"] = f"watchdog: not finished after {LEG_TIMEOUT_S} s (a rank failed inside a collective?)"
            line["strategies"], line["emulated_rank_ms"] = strategies, emulated
            emit()
            sys.stdout.flush()
            os._exit(0)

        watchdog = threading.Timer(LEG_TIMEOUT_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        sH_, sW_ = round(scale * H), round(scale * W)
        ny_, nx_ = -(-H // int(cfg.block_matching.tuning.tile_size)), -(-W // int(cfg.block_matching.tuning.tile_size))

        def rccl_bytes(strategy):
            """Payload this rank hands to its collective(s) per burst (what crosses xGMI is (G - 1) / G of it per rank)."""
            if strategy == "rows":  # all-gather of the flow fields: every rank contributes its rounds' slots
                rounds = -(-(NF - 1) // world)
                return {"all_gather_in_bytes": rounds * ny_ * nx_ * 2 * 4, "all_gather_out_bytes": world * rounds * ny_ * nx_ * 2 * 4}
            rows_ = hdist.slab_rows(sH_, world)  # reduce-scatter of the packed accumulators [world, 2, rows, sW, 3]
            return {"reduce_scatter_in_bytes": world * 2 * rows_ * sW_ * 3 * 4, "reduce_scatter_out_bytes": 2 * rows_ * sW_ * 3 * 4}

        def max_over_ranks(v):
            t = torch.tensor([float(v)], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        def compute_only_ms(eng, strategy):
            """max over ranks of the rank's step with its collectives left out (HipEngine with graphs only)."""
            fn = None
            if on_gpu and strategy == "rows" and getattr(eng, "_plans", None):
                plan = next(iter(eng._plans.values()))
                fn = plan.replay_compute_only
            elif on_gpu and strategy == "reduce":
                rows_ = hdist.slab_rows(sH_, world)
                bounds_ = hdist.slab_bounds(sH_, world)
                mine = [comp[i] for i in hdist.shard_indices(NF - 1, rank, world)]
                red = torch.zeros((2, rows_, sW_, 3), dtype=torch.float32, device=dev)

                def fn():
                    acc, _, ref_dev, ref_covs = eng.partial(ref, mine, bounds_, rows_)
                    red.copy_(acc[rank])  # stand-in for the reduce-scatter
                    return eng.finish_rows(red, bounds_[rank], bounds_[rank + 1], ref_dev, ref_covs) if bounds_[rank + 1] > bounds_[rank] else None
            ok = torch.tensor([1 if fn is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if not int(ok.item()):
                return None
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            return max_over_ranks((time.perf_counter() - t0) / args.steps * 1e3)

        try:
            strategies = {args.strategy: {"ms_per_step": round(ms_per_step, 3), "value": round(value, 2), "headline": True,
                                          "rccl_bytes_per_rank": rccl_bytes(args.strategy)}}
            if on_gpu:
                strategies[args.strategy]["compute_only_max_rank_ms"] = compute_only_ms(engine, args.strategy)
            other = "reduce" if args.strategy == "rows" else "rows"
            eng_o = engine_cls(cfg)
            fn_o = lambda: hdist.main_sharded(ref, comp, cfg, engine=eng_o, gather=args.gather, strategy=other,  # noqa: E731
                                              max_flow=args.max_flow)[0]
            ms_o = timed(fn_o, args.steps, max(3, min(args.warmup, 5)) if on_gpu else min(args.warmup, 1))
            strategies[other] = {"ms_per_step": round(ms_o, 3), "value": round(out_pix / (ms_o * 1e-3) / 1e6, 2), "headline": False,
                                 "rccl_bytes_per_rank": rccl_bytes(other)}
            if on_gpu:
                strategies[other]["compute_only_max_rank_ms"] = compute_only_ms(eng_o, other)
            del eng_o
            # the committed single-GPU emulation of this rank count (tools/debug/emulate_ranks.py: every rank's graphs replayed
            # alone on ONE MI355X, collectives left out) for the same workload, if there is one
            emulated = {}
            for fname in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
                if "emulate_ranks" in fname and fname.endswith(".jsonl") and "staged" not in fname:
                    for ln in open(os.path.join(ROOT, "profiles", fname)):
                        rec = json.loads(ln)
                        if rec.get("workload") == f"{H}x{W}x{NF} x{scale}" and rec.get("world") == world and \
                                rec["strategy"] not in emulated:
                            emulated[rec["strategy"]] = {"max_rank_ms": rec["max_rank_ms"], "mean_rank_ms": rec["mean_rank_ms"],
                                                         "source": f"profiles/{fname}"}
            emulated = emulated or None
        except Exception as e:  # noqa: BLE001
            errors["strategies_leg"] = f"{type(e).__name__}: {e}"
        watchdog.cancel()

    # ---- descriptive fields + the whole-step roofline (nothing below this block can lose the measurement) -----------
    sb = step_bytes(NF - 1, P, S) / world  # per rank: every rank's kernels cover 1 / world of the burst's work
    line.update({
        "config": {"workload": f"{H}x{W} Bayer burst, {NF} frames, x{scale} SR, Ts={cfg.block_matching.tuning.tile_size}, "
                               f"metrics={cfg.block_matching.tuning.metrics}, robustness on, frames resident in HBM "
                               f"(`value`, `vs_baseline`: device-resident scope; `value_reference_scope`, "
                               f"`vs_baseline_reference_scope`: the reference's timer scope, frames start as host float32 "
                               f"arrays, super_resolution.py:103-195)",
                   "l1_semantics": "intended (SAD argmin; the reference's level-0 L1 kernel is undefined behaviour "
                                   "upstream, SURVEY.md App. A D1: block_matching.py:168-180; oracle-defined, unpinned)",
                   "parallelism": ((f"{world} ranks, strategy rows: alignment frame-parallel, all-gather of flows, merge "
                                    f"row-parallel" if args.strategy == "rows" else
                                    f"{world} ranks, strategy reduce: frames one per rank, reduce-scatter of the float32 "
                                    f"accumulators over row slabs") +
                                   f", output {'gathered to rank 0' if args.gather else 'sharded by rows'}")
                   if world > 1 else "single GPU"},
        "strategy": args.strategy if world > 1 else None,
        "launch": (("HIP graph replay: the step is captured once (stream capture incl. the frame pipeline's side "
                    "streams) and replayed with one launch per burst; every kernel runs on every step" if world == 1
                    else ("HIP graph replay of the per-rank steps (alignment of the rank's frames; robustness + "
                          "kernels + merge of its rows), the all-gather of the flow fields between them"
                          if args.strategy == "rows" else
                          "HIP graph replay of the two per-rank steps (the rank's frames through the whole chain into "
                          "accumulators; reference frame + normalisation of its slab), the reduce-scatter between them"))
                   if graphed else "one launch per kernel from Python"),
        "ms_per_step_eager": round(ms_eager, 3) if ms_eager else None,
        "rccl_ranks": dist.get_world_size() if world > 1 else 1,
        "ranks_agree": (dist.get_world_size() == world == args.gpus) if world > 1 else True,  # launcher, --gpus and the group
        "strategies": strategies, "emulated_rank_ms": emulated,
        "backend": (args.backend if world > 1 else None),
        "engine": "HipEngine (libhhsr_hip.so)" if on_gpu else f"{args.engine} (launch-plumbing test, not a measurement)",
        "step_roofline": {"algorithmic_bytes": sb, "achieved": round(sb / (ms_per_step * 1e-3) / 1e9, 1),
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(sb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "whole step, PER RANK: algorithmic bytes of every kernel of the fused pipeline "
                                  "(bench.step_bytes) / ranks over the step time against ONE GPU's HBM peak; its kernels are "
                                  "VALU- or latency-bound (profiles/*_kernel_bottlenecks.md), none is HBM-bound"},
        "reference_published": "48 MP in < 4 s (>= 12 output Mpix/s) on an RTX 3090 for a 20-frame burst (README.md:10)",
    })

    # ---- dominant-kernel roofline -------------------------------------------------------------------------------------
    roof = None
    try:
        step_ms = [e0.elapsed_time(e1) for e0, e1, nfr in ev if nfr > 0]
        if step_ms:
            per_step = len(step_ms) // ev_steps if ev_steps and len(step_ms) % ev_steps == 0 else 0
            if per_step:  # launches of one step summed, then the MEDIAN over the steps
                sums = [float(np.sum(step_ms[i * per_step:(i + 1) * per_step])) for i in range(ev_steps)]
                avg_ms = float(np.median(sums))
                launch_spread = (min(sums), max(sums))
            else:
                avg_ms = float(np.sum(step_ms)) / ev_steps
                launch_spread = None
            # whole-image launch on one GPU; on N ranks each launch covers 1/N of the output rows (+ halo rows of input)
            nbytes = merge_burst_bytes(NF - 1, P, S) / world
            achieved = nbytes / (avg_ms * 1e-3) / 1e9
            # the kernel hhsr_merge_burst picks for this scale (csrc/hhsr_merge*.hip: x2 and x3 have wave-per-parity-class
            # kernels, other integer scales the tile kernel, non-integer scales the generic one)
            kernel = ("k_merge_x2" if float(scale) == 2.0 else "k_merge_xs<3>" if float(scale) == 3.0 and W % 4 == 0
                      else "k_merge_burst_tile" if float(scale).is_integer() else "k_merge_burst")
            traffic, valu, pmc_note = pmc_lookup(scale, H, W, NF, avg_ms) if world == 1 else (None, None, None)
            roof = {"kernel": f"{kernel} (hhsr_merge_burst)", "bound": "valu", "achieved": round(achieved, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "avg_launch_ms_is": "median over the eager steps",
                    "launch_ms_min_max": [round(v, 4) for v in launch_spread] if launch_spread else None,
                    "bytes_per_launch": nbytes, "valu_issue": valu,
                    "frac_valu": valu["frac"] if valu else None, "frac_hbm": round(achieved / HBM_PEAK_GBS, 4),
                    "note": "the fused burst merge keeps the accumulators in registers: ~100 flop per byte, bound by VALU "
                            "issue (frac_valu), not by HBM; achieved / peak / frac are the HBM figures the contract asks for"
                            + ("; launch duration from HIP events around the launch in eager steps of the same workload "
                               "right after the timed graph replays" if graphed else "")
                            + ("; " + pmc_note if pmc_note else "")}

    except Exception as e:  # noqa: BLE001
        errors["roofline"] = f"{type(e).__name__}: {e}"
    line["roofline"] = roof

    # ---- H2D-inclusive legs: the reference's scope (frames are host arrays when the timer starts); N = 1 only --------
    h2d = None
    if on_gpu and not args.no_h2d and world == 1:
        import copy

        steps_h = max(3, args.steps // 2)
        nbytes_h = 4.0 * H * W * NF
        PCIE_GBS = 63.0  # PCIe Gen5 x16 spec, one direction
        floor = nbytes_h / (PCIE_GBS * 1e9) * 1e3

        def host_leg(r_h, c_h, cfg_h):
            """A burst starting in host memory: one engine kept across the bursts (graph.HostBurstRunner: eager uploads
            into static staging, per-chunk HIP graphs of the front end, one merge graph)."""
            eng_h = engine_cls(cfg_h)
            fn = lambda: hdist.main_sharded(r_h, c_h, cfg_h, engine=eng_h, gather=args.gather, strategy=args.strategy,  # noqa: E731
                                            max_flow=args.max_flow)[0]
            # eager, capture, then replays until the copy path is at speed: after a compute-only phase (the legs above) the
            # first ~100 ms of H2D copies run at HALF rate (rocprofv3 --memory-copy-trace: 1.6 instead of 0.85 ms per 48 MB,
            # tools/debug/copy_stats.py) — link / DMA clocks ramping; the timed steps are the steady state of a serving loop
            ms = timed(fn, steps_h, 10)
            runner = getattr(eng_h, "_host", None)
            graphs = bool(runner is not None and not runner.disabled and any(s != "seen" for s in runner.states.values()))
            del eng_h
            return ms, graphs

        ref_h = comp_h = None
        try:
            ref_h = ref.cpu().pin_memory()
            comp_h = [comp[i].cpu().pin_memory() for i in range(NF - 1)]  # one pinned float32 array per frame
            ms_h, g_h = host_leg(ref_h, comp_h, cfg)
            h2d = {"value_incl_h2d": round(out_pix / (ms_h * 1e-3) / 1e6, 2), "ms_per_step_incl_h2d": round(ms_h, 3),
                   "steps": steps_h, "host_bytes_per_step": nbytes_h, "pcie_floor_ms": round(floor, 2),
                   "pcie_floor_frac": round(floor / ms_h, 3), "graphs": g_h,
                   "note": "frames start as pinned host float32 (the reference's timer scope, super_resolution.py:103-195): "
                           "uploads are eager hipMemcpyAsync calls back to back on one upload stream into static staging "
                           "buffers, the kernels replay as per-chunk HIP graphs that wait for their frames' copies "
                           "(graph.HostBurstRunner); pcie_floor = bytes / 63 GB/s (PCIe Gen5 x16 spec)"}
            # the reference-scope figure leads the line next to `value`
            line.update(value_reference_scope=h2d["value_incl_h2d"], ms_per_step_reference_scope=h2d["ms_per_step_incl_h2d"],
                        vs_baseline_reference_scope=round(h2d["value_incl_h2d"] / 12.0, 2) if headline else None,
                        value_incl_h2d=h2d["value_incl_h2d"], ms_per_step_incl_h2d=h2d["ms_per_step_incl_h2d"],
                        vs_baseline_incl_h2d=round(h2d["value_incl_h2d"] / 12.0, 2) if headline else None, h2d=h2d)
        except Exception as e:  # noqa: BLE001
            errors["h2d_f32_leg"] = f"{type(e).__name__}: {e}"
        if h2d is not None:
            try:
                # the same scope with the frames as the sensor's uint16 counts (what a DNG holds; the reference converts them
                # to float32 on the host, utils_dng.py:149-160): half the PCIe bytes, normalised on the device per frame
                black, white = 64.0, 1023.0
                to_counts = lambda t: torch.from_numpy(np.clip(np.rint(t.numpy() * (white - black) + black), 0, white)  # noqa: E731
                                                       .astype(np.uint16)).pin_memory()
                ref16, comp16 = to_counts(ref_h), [to_counts(c) for c in comp_h]
                cfg16 = copy.deepcopy(cfg)
                cfg16.hip = dict(cfg16.get("hip", None) or {}, raw_norm={"black_levels": [black] * 3, "white_level": white})
                ms_16, g_16 = host_leg(ref16, comp16, cfg16)
                h2d.update(value_incl_h2d_u16=round(out_pix / (ms_16 * 1e-3) / 1e6, 2), ms_per_step_incl_h2d_u16=round(ms_16, 3),
                           pcie_floor_ms_u16=round(floor / 2, 2), pcie_floor_frac_u16=round(floor / 2 / ms_16, 3), graphs_u16=g_16,
                           note_u16="frames start as pinned host uint16 sensor counts (10-bit, black 64): uploaded as "
                                    "counts, normalised on the device frame by frame (hhsr_normalize_raw_u16)")
                del ref16, comp16
            except Exception as e:  # noqa: BLE001
                errors["h2d_u16_leg"] = f"{type(e).__name__}: {e}"
            try:
                # the reference's literal call: main(ref, comp, config) with plain (pageable) NumPy float32 arrays
                from handheld_super_resolution.graph import HostBurstRunner

                ref_np, comp_np = ref_h.numpy().copy(), np.stack([c.numpy() for c in comp_h])
                cfg_np = copy.deepcopy(cfg)
                fn_np = lambda: hsr.main(ref_np, comp_np, cfg_np)[0]  # noqa: E731
                ms_np = timed(fn_np, steps_h, 10)
                h2d.update(value_numpy_pageable=round(out_pix / (ms_np * 1e-3) / 1e6, 2), ms_per_step_numpy_pageable=round(ms_np, 3),
                           note_numpy=f"hsr.main(ref, comp, config) on pageable NumPy float32 arrays, the same config object "
                                      f"call after call: {HostBurstRunner.COPY_THREADS} host threads copy the frames into "
                                      f"page-locked staging while earlier frames cross PCIe; fresh result tensor per call "
                                      f"(one device copy)")
                del ref_np, comp_np
            except Exception as e:  # noqa: BLE001
                errors["h2d_numpy_leg"] = f"{type(e).__name__}: {e}"
        del ref_h, comp_h

    # ---- the reference's weight typing, timed: Numba types the merge's coordinates and weights as float64
    # (merge.py:319-426); the timed step above computes them in float32 within the stated tolerance.
    # config.hip.weight_fp64 runs the float64 chain on every pixel (the generic per-pixel kernel): what reference-typed
    # arithmetic costs on this chip.  N = 1, guarded like the other legs.
    if on_gpu and world == 1 and not args.no_h2d:
        import copy

        try:
            cfg64 = copy.deepcopy(cfg)
            cfg64.hip = dict(cfg64.get("hip", None) or {}, weight_fp64=True)
            eng64 = engine_cls(cfg64)
            fn64 = lambda: hdist.main_sharded(ref, comp, cfg64, engine=eng64, gather=args.gather)[0]  # noqa: E731
            ms64 = timed(fn64, max(3, args.steps // 4), 3)
            line.update(ms_per_step_weight_fp64=round(ms64, 3), value_weight_fp64=round(out_pix / (ms64 * 1e-3) / 1e6, 2))
            del eng64
        except Exception as e:  # noqa: BLE001
            errors["weight_fp64_leg"] = f"{type(e).__name__}: {e}"

    # ---- the accumulated-robustness merge denoiser on (merge.py:223-228; off in configs/default.yaml): the comp frames through
    # the same fused merge, the float64 robustness sum in one pass (hhsr_rob_sum), then merge_ref with its overwrite / widen
    # rules + divide.  N = 1, guarded.
    if on_gpu and world == 1 and not args.no_h2d:
        import copy

        try:
            cfgd = copy.deepcopy(cfg)
            cfgd.accumulated_robustness_denoiser.enabled = True
            cfgd.accumulated_robustness_denoiser.merge.enabled = True
            engd = engine_cls(cfgd)
            fnd = lambda: hdist.main_sharded(ref, comp, cfgd, engine=engd, gather=args.gather)[0]  # noqa: E731
            msd = timed(fnd, max(3, args.steps // 2), 3)
            line.update(ms_per_step_denoiser=round(msd, 3), value_denoiser=round(out_pix / (msd * 1e-3) / 1e6, 2))
            del engd
        except Exception as e:  # noqa: BLE001
            errors["denoiser_leg"] = f"{type(e).__name__}: {e}"

    # ---- the C5 geometry on ONE GPU (48 MP x 20 frames x3 -> 432 MP; BASELINE.json configs[4] is this workload over 8 GPUs):
    # the x3 merge kernel's numbers in the driver's own record.  Optional, guarded, N = 1, headline run only.
    if on_gpu and world == 1 and headline and not args.no_c5 and not args.no_h2d:
        import copy

        try:
            H5, W5, NF5, sc5 = 6000, 8000, 20, 3
            ref5, comp5, _ = synth.make_burst_torch(H5, W5, NF5, dev, seed=4321)
            cfg5 = hsr.default_config()
            cfg5.verbose = 0
            cfg5.scale = sc5
            cfg5.hip = {"graph": not args.no_graph}
            hsr.prepare_config(cfg5, np.full((H5, W5), float(ref5.mean()), np.float32), synth.ALPHA_ISO100, synth.BETA_ISO100,
                               [[0, 1], [1, 2]], [1.0, 1.0, 1.0])
            eng5 = engine_cls(cfg5)
            fn5 = lambda: hdist.main_sharded(ref5, comp5, cfg5, engine=eng5)[0]  # noqa: E731
            blocks5, clocks5 = timed_reps(fn5, 3, 3, 3)
            sp5 = spread(blocks5)
            del eng5
            out5 = sc5 * H5 * sc5 * W5
            c5 = {"workload": f"{H5}x{W5} Bayer burst, {NF5} frames, x{sc5} SR, frames resident in HBM, one GPU",
                  "ms_per_step": round(sp5["median"], 3), "ms_per_step_min": round(sp5["min"], 3),
                  "ms_per_step_max": round(sp5["max"], 3), "steps": 3, "reps": len(blocks5),
                  "value": round(out5 / (sp5["median"] * 1e-3) / 1e6, 2), "unit": "Mpix/s",
                  "sclk_mhz": round(spread(clocks5)["median"], 1) if clocks5 else None, "roofline": None}
            line["c5_one_gpu"] = c5
            # the x3 merge kernel's launch duration: HIP events around the launch in eager steps of the same workload
            cfg5e = copy.deepcopy(cfg5)
            cfg5e.hip = dict(cfg5e.hip, graph=False)
            eng5e = engine_cls(cfg5e)
            fn5e = lambda: hdist.main_sharded(ref5, comp5, cfg5e, engine=eng5e)[0]  # noqa: E731
            for _ in range(2):
                fn5e()
            barrier()
            ev5 = []
            timed_call.sink, timed_call.on = ev5, True
            timed(fn5e, 3, 0)
            timed_call.on, timed_call.sink = False, ev
            del eng5e
            ms5 = [e0.elapsed_time(e1) for e0, e1, nfr in ev5 if nfr > 0]
            if ms5 and len(ms5) % 3 == 0:
                k5 = len(ms5) // 3
                launch5 = float(np.median([float(np.sum(ms5[i * k5:(i + 1) * k5])) for i in range(3)]))
                nb5 = merge_burst_bytes(NF5 - 1, H5 * W5, float(sc5) ** 2)
                ach5 = nb5 / (launch5 * 1e-3) / 1e9
                tr5, valu5, note5 = pmc_lookup(sc5, H5, W5, NF5, launch5)
                c5["roofline"] = {"kernel": "k_merge_xs<3> (hhsr_merge_burst)", "bound": "valu", "achieved": round(ach5, 1),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach5 / HBM_PEAK_GBS, 4), "traffic": tr5,
                                  "avg_launch_ms": round(launch5, 4), "bytes_per_launch": nb5, "valu_issue": valu5,
                                  "frac_valu": valu5["frac"] if valu5 else None, "note": note5}
            del ref5, comp5
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            timed_call.on, timed_call.sink = False, ev
            errors["c5_leg"] = f"{type(e).__name__}: {e}"

    # ---- CPU baseline (all host cores) + parity with attribution, on a crop of the same burst ----------------------
    cpu, parity = None, None
    if rank == 0 and world == 1 and on_gpu and not args.no_cpu_baseline:
        try:
            import oracle

            c = min(args.cpu_crop, H, W)
            c -= c % 32
            # as many disjoint crops of the same burst as the host has cores for (one worker process per comp frame and crop),
            # all processed at the same time; crop 0 (the centre) doubles as the parity sample
            cores = args.cpu_cores or oracle.available_cores()  # (affinity mask capped by the cgroup CPU quota)
            k_crops = max(1, min(cores // max(1, NF - 1), (H // c) * (W // c)))
            y0, x0 = ((H - c) // 64) * 32, ((W - c) // 64) * 32
            origins = [(y0, x0)] + [(gy * c, gx * c) for gy in range(H // c) for gx in range(W // c)][:k_crops - 1]
            crops = [(ref[y:y + c, x:x + c].cpu().numpy(), comp[:, y:y + c, x:x + c].cpu().numpy()) for y, x in origins]
            ref_c, comp_c = crops[0]
            cap = {}
            want, _, tc, cores_used = oracle.throughput_all_cores(crops, cfg, cores=cores, capture=cap)
            cpu = {"value": round(len(crops) * round(scale * c) ** 2 / tc / 1e6, 4), "unit": "output Mpix/s", "cores": cores_used,
                   "host_cores": os.cpu_count(), "usable_cores": oracle.available_cores(), "kind": "port",
                   "sample": f"{len(crops)} crop(s) of {c}x{c} of the same burst processed concurrently, all {NF} frames each, "
                             f"x{scale}, NumPy oracle (golden-pinned port; the reference has no CPU path), {cores_used} worker "
                             f"processes = every core the container may use (logical CPUs capped by the cgroup CPU quota; "
                             f"{os.cpu_count()} logical CPUs visible), {tc:.1f} s wall"}

            def diff(cfg_run):
                got = hsr.main(ref_c, comp_c, cfg_run)
                o = got[0].cpu().numpy()
                with np.errstate(all="ignore"):
                    d = np.abs(o.astype(np.float64) - want.astype(np.float64))
                return o, d, got[1]

            cfg_d = cfg.copy()
            cfg_d.debug = True
            got, dabs, dbg = diff(cfg_d)
            fin = np.isfinite(dabs)
            # tiles whose flow differs by more than ICA noise: a block-matching decision flipped (float32 near-tie)
            gflow, oflow = np.stack(dbg["flow"]), np.stack(cap["flow"])
            flipped = np.abs(gflow - oflow).max(-1) > 0.05
            cfg_i = cfg.copy()
            cfg_i.hip = dict(cfg.get("hip", None) or {}, inject_flows=[f for f in cap["flow"]])
            _, dinj, _ = diff(cfg_i)
            fin_i = np.isfinite(dinj)
            # ... and the other side (round 5): the ORACLE's robustness + kernels + merge on HIP's flows (C form of the
            # accumulation, oracle.cfast) against HIP's own-flow image — identical flows on both sides again
            want_h, _, _ = oracle.main_parallel(ref_c, comp_c, cfg, workers=min(cores, NF - 1), fast=True, flows=list(gflow))
            with np.errstate(all="ignore"):
                dh = np.abs(got.astype(np.float64) - want_h.astype(np.float64))
            fin_h = np.isfinite(dh)
            parity = {"max_abs_diff": float(dabs[fin].max()), "p999_abs_diff": float(np.percentile(dabs[fin], 99.9)),
                      "frac_above_1e-4": float((dabs[fin] > 1e-4).mean()),
                      "nan_mismatch": int((np.isnan(got) != np.isnan(want)).sum()),
                      "flipped_tiles": int(flipped.sum()), "tiles": int(flipped.size),
                      "max_flow_diff_unflipped_px": float(np.abs(gflow - oflow).max(-1)[~flipped].max()),
                      "max_abs_diff_oracle_flows_injected": float(dinj[fin_i].max()),
                      "max_abs_diff_vs_oracle_on_hip_flows": float(dh[fin_h].max()),
                      "nan_mismatch_vs_oracle_on_hip_flows": int((np.isnan(got) != np.isnan(want_h)).sum()),
                      "vs": "oracle (golden-pinned port)", "sample": f"{c}x{c} crop, {NF} frames, x{scale}",
                      "note": "flipped_tiles = tiles (over all comp frames) whose flow differs from the oracle's by > 0.05 px: "
                              "float32 near-ties of one block-matching decision; with the oracle's flow fields injected "
                              "(config.hip.inject_flows) the remaining difference is the arithmetic of robustness + kernels + merge; "
                              "max_abs_diff_vs_oracle_on_hip_flows is the same comparison the other way round: the oracle's "
                              "robustness + kernels + merge run on HIP's flow fields (the two-sided contract of "
                              "tests/test_fuzz_parity.py)"}

        except Exception as e:  # noqa: BLE001
            errors["cpu_parity_leg"] = f"{type(e).__name__}: {e}"
    line.update(cpu_baseline=cpu, parity=parity)
    emit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

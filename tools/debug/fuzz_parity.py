"""Randomised end-to-end parity sweep (not part of the test suite: ~15 s of NumPy oracle per case).
   python tools/debug/fuzz_parity.py [n_cases] [seed]
Random sizes (also odd multiples of 2), frame counts, scales, CFA patterns, white balances, tile sizes, kernels,
robustness on / off, denoiser on / off; HIP main() vs oracle.main(): flipped block-matching tiles, largest differences of
flow / r / image outside their footprint."""
import sys
import numpy as np
import torch
sys.path.insert(0, "handheld-multi-frame-super-resolution_amd"); sys.path.insert(0, "."); sys.path.insert(0, "tests")
import oracle
import handheld_super_resolution as hsr
from handheld_super_resolution import synthetic as synth
from helpers import base_config, flipped_tiles, footprint

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else None  # run (and dissect) just this case of the sequence
CFAS = [((0, 1), (1, 2)), ((2, 1), (1, 0)), ((1, 0), (2, 1)), ((1, 2), (0, 1))]
worst = 0.0
for k in range(n_cases):
    ts = int(rng.choice([16, 16, 32]))
    lo, hi = (576, 760) if ts == 16 else (384, 640)  # (default pyramid factors need >= 36 tiles of 16; ts 32: factors 2)
    H = int(rng.integers(lo, hi)) // 2 * 2
    W = int(rng.integers(lo, hi + 64)) // 2 * 2
    nf = int(rng.integers(2, 5))
    scale = [1, 1.5, 2, 2, 3][int(rng.integers(0, 5))]
    cfa = CFAS[int(rng.integers(0, 4))]
    wb = (float(rng.uniform(1.0, 2.2)), 1.0, float(rng.uniform(1.0, 2.0))) if rng.random() < 0.7 else (1.0, 1.0, 1.0)
    iso = rng.random() < 0.2
    rob = rng.random() < 0.85
    den = rob and rng.random() < 0.25
    seed = int(rng.integers(0, 10000))
    max_shift, occ = float(rng.uniform(0.5, 3.5)), bool(rng.random() < 0.5)
    if only is not None and k != only:
        continue

    def cfg_fn():
        c = base_config(ts=ts, scale=scale, snr=30.0 if ts == 16 else 18.0)
        if ts == 32:
            c.block_matching.tuning.factors = [1, 2, 2, 2]
        c.exif = {"cfa_pattern": [list(r) for r in cfa], "iso": 100, "white_balance": list(wb)}
        c.robustness.enabled = rob
        c.robustness.save_mask = rob
        if iso:
            c.merging.kernel = "iso"
        if den:
            c.accumulated_robustness_denoiser.enabled = True
            c.accumulated_robustness_denoiser.merge.enabled = True
        return c

    ref, comp, _ = synth.make_burst(H, W, nf, seed=seed, max_shift=max_shift, occluder=occ, cfa=cfa, wb=wb)
    cap = {}
    want, wdbg = oracle.main(ref, comp, cfg_fn(), capture=cap)  # (in-process: a failing pool initialiser respawns forever)
    c = cfg_fn()
    c.debug = True
    out, dbg = hsr.main(ref, comp, c)
    o = out.cpu().numpy()
    gflow, oflow = np.stack(dbg["flow"]), np.stack(cap["flow"])
    fl = flipped_tiles(gflow, oflow)
    mask = footprint(fl, ts, (H, W), scale)
    d = np.abs(o - want)
    nanmis = int((np.isnan(o) != np.isnan(want)).sum())
    d = np.where(np.isnan(d), 0, d)
    dout = float(np.where(mask[..., None], 0, d).max())
    dflow = float(np.abs(gflow - oflow).max(-1)[~fl].max()) if (~fl).any() else 0.0
    dr = 0.0
    if rob:
        m1 = footprint(fl, ts, (H, W), 1.0)
        dr = float(np.where(m1[None], 0, np.abs(np.stack(dbg["robustness"]) - np.stack(cap["r"]))).max())
    worst = max(worst, dout)
    flag = "  <-- ABOVE 1e-4" if (dout > 1e-4 or dr > 1e-4 or nanmis) else ""
    print(f"case {k:2d}: {H}x{W} x{nf} s={scale} ts={ts} cfa={cfa} wb={tuple(round(v, 2) for v in wb)} iso={iso} rob={rob} den={den} "
          f"seed={seed}: flipped {int(fl.sum())}, flow {dflow:.1e}, r {dr:.1e}, image {dout:.2e}, nan mismatches {nanmis}{flag}",
          flush=True)
    if only is not None:  # where, and what removes it
        idx = np.argwhere(d > 0.5 * d.max())
        for y, x, ch in idx[:10]:
            print("   ", (int(y), int(x), int(ch)), float(o[y, x, ch]), float(want[y, x, ch]), float(d[y, x, ch]))
        df = np.abs(gflow - oflow).max(-1)
        kk = np.unravel_index(df.argmax(), df.shape)
        print("    max flow diff at (frame, ty, tx)", tuple(int(v) for v in kk), float(df.max()), gflow[kk], oflow[kk])
        for name, hip in (("float64 weight chain", {"weight_fp64": True}), ("oracle flows injected", {"inject_flows": [f for f in oflow]})):
            c2 = cfg_fn()
            c2.hip = hip
            o2 = hsr.main(ref, comp, c2)[0].cpu().numpy()
            d2 = np.abs(o2 - want)
            d2 = np.where(np.isnan(d2), 0, d2)
            print(f"    {name}: max {float(d2.max()):.2e}")
            if d2.max() > 1e-4:
                for y, x, ch in np.argwhere(d2 > 0.5 * d2.max())[:8]:
                    print("       ", (int(y), int(x), int(ch)), float(o2[y, x, ch]), float(want[y, x, ch]), "default path:", float(o[y, x, ch]))
                y, x, ch = (int(v) for v in np.argwhere(d2 > 0.5 * d2.max())[0])
                lry, lrx = (y + 0.5) / scale, (x + 0.5) / scale
                print(f"        LR position ({lry}, {lrx}); flows of its tile ({int(lry) // ts}, {int(lrx) // ts}) per frame (x, y):")
                for f_ in range(gflow.shape[0]):
                    print("          frame", f_, "gpu", gflow[f_, int(lry) // ts, int(lrx) // ts].tolist(), "oracle",
                          oflow[f_, int(lry) // ts, int(lrx) // ts].tolist(),
                          "r gpu/oracle at pixel", float(np.stack(dbg["robustness"])[f_, int(lry), int(lrx)]), float(np.stack(cap["r"])[f_, int(lry), int(lrx)]))
                c3 = cfg_fn()
                c3.hip = dict(hip, merge_kernel="generic")
                o3 = hsr.main(ref, comp, c3)[0].cpu().numpy()
                d3 = np.where(np.isnan(np.abs(o3 - want)), 0, np.abs(o3 - want))
                print(f"        same with the generic merge kernel: max {float(d3.max()):.2e}")
                c4 = cfg_fn()
                c4.hip = dict(hip)
                c4.debug = True
                o4, dbg4 = hsr.main(ref, comp, c4)
                o4 = o4.cpu().numpy()
                d4 = np.where(np.isnan(np.abs(o4 - want)), 0, np.abs(o4 - want))
                dr4 = np.abs(np.stack(dbg4["robustness"]) - np.stack(cap["r"]))
                print(f"        same on the per-frame (debug) path: max {float(d4.max()):.2e}; r max {float(dr4.max()):.2e} at "
                      f"{tuple(int(v) for v in np.unravel_index(dr4.argmax(), dr4.shape))}")
                r4, ro = np.stack(dbg4["robustness"]), np.stack(cap["r"])
                for f_ in range(r4.shape[0]):
                    print(f"        frame {f_}: r at the pixel's LR position gpu {float(r4[f_, int(lry), int(lrx)])!r} oracle "
                          f"{float(ro[f_, int(lry), int(lrx)])!r}; image at the pixel gpu {float(o4[y, x, ch])!r} oracle {float(want[y, x, ch])!r}")
                # contribution test: the same burst without the comp frames (reference frame only)
                c5 = cfg_fn()
                o5 = hsr.main(ref, comp[:0], c5)[0].cpu().numpy()
                w5, _ = oracle.main(ref, comp[:0], cfg_fn())
                print(f"        reference frame alone at the pixel: gpu {float(o5[y, x, ch])!r} oracle {float(w5[y, x, ch])!r}")
print("worst image difference outside flipped-tile footprints:", worst)

"""Randomised end-to-end parity sweep as a test: HIP main() against oracle.main() on 64 random bursts —
sizes (also odd multiples of 2), 2-4 frames, scales 1 / 1.5 / 2 / 3, the four Bayer patterns, white balances, tile sizes
16 / 32, iso kernel, robustness and merge denoiser on / off, moving occluders, level-0 metric L2 / L1 / L1_ref_effective.

Asserted per case (helpers.fuzz_verdict and its constants FLIP_PX, CLUSTER, MAX_ICA_TILES, MAX_OUTLIERS, MAX_OUTLIER,
NUM_ERR; every case also runs with the oracle's flow fields injected, config.hip.inject_flows — that run
exercises kernels + robustness + merge on identical geometry, the flow comparison exercises the alignment):
  * identical NaN pattern (and equal infinities);
  * flow <= 1e-4 px (measured <= 9.9e-5) on every tile EXCEPT
      - the tiles under ONE flipped block-matching decision per case: a float32 near-tie somewhere in the pyramid, which
        all finest-level tiles under that coarser tile inherit — the tiles whose flow differs by > FLIP_PX = 1e-3 px
        (measured 0.04 - 0.11 px) must lie in one frame inside a bounding box of CLUSTER x CLUSTER tiles (measured: a
        single tile in the 64 cases, 2 x 2 blocks in two of the 576 held-out cases), at most FLIPPED_PER_BATCH per batch;
      - at most MAX_ICA_TILES tiles per case between 1e-4 and 1e-3 px: ill-conditioned Lucas-Kanade systems at a moving
        occluder, where three ICA iterations amplify the float32 noise of the gradient sums (measured: 1, 1 and 10 tiles,
        <= 3.4e-4 px, in 3 of the 576 held-out cases, none in the 64);
    robustness r <= 1e-4 outside the footprint of all those tiles;
  * oracle flows injected: image <= 1e-4 wherever every frame is fully accepted (r = 1 in the 5 x 5 raw-pixel
    neighbourhood; with the robustness off: everywhere); where some frame is being rejected at most two raw
    pixels' worth of isolated values per case (2 x 3 x ceil(scale)^2: one raw pixel is scale^2 output pixels x 3
    channels), each <= MAX_OUTLIER or, where the value's accumulated weight den is smaller than NUM_ERR / MAX_OUTLIER,
    <= NUM_ERR / den (case 101.9: 0.104 at den = 3.4e-7 — every outlier of set 100 has |d out| x den <= 8.9e-8, float32
    rounding of num and den themselves) — mechanism (a) below.  Measured: the 64 cases: 63 <= 9.6e-5, one with 2 values at
    1.01e-4; nine held-out sets of 64 (HHSR_FUZZ_BATCHES=10:22,11:22,12:20 ... 90:22,91:22,92:20): 546 of 576 <= 1e-4,
    30 cases with 1 - 4 values each (18 and 22 — one raw pixel — in two cases at scale 3) between 1.1e-4 and 2.3e-3.
    (Earlier forms of this assertion — each value <= 3e-4; values above 1.05e-4 only in tiles displaced by > 30 px; <= 16
    values per case — were calibrated on the 64 fixed cases and FAILED on held-out sets: r in its transition band is
    what the exceptions have in common, not a diverged alignment.)
  * own flows: image <= 1e-4 outside the footprint of a flipped tile EXCEPT
      (a) isolated pixels where some frame is being rejected: at most MAX_OUTLIERS values per case, each <= MAX_OUTLIER;
      (b) flow-sensitive pixels — pixels that agree (<= 1e-4) once the oracle's flows are injected, i.e. whose whole
          difference comes from the <= 1e-4 px by which the flows differ: at most two tiles' worth per case.

Why those pixels exist — conditioning of the reference algorithm, not arithmetic differences:
(a) where a frame is being rejected, r sits in its transition band (1e-5 ... 1e-3) at some taps: R = S e - t cancels to
1e-4 of its operands, and where the frame's sample and the reference sample weigh about the same the normalised value
(a w_ref + b r w) / (w_ref + r w) moves by (b - a) / 4r per unit of r — HIP and oracle r differing by 8e-7 around r =
5.9e-4 is 1.5e-4 in the image (case 32.7, no occluder, two frames); under a moving occluder or a diverged alignment the
same happens with larger (b - a) (6.3e-4, case 10.0).  The reference's own float32 buffers carry the same rounding noise.
(b) the 3 x 3 tap window is centred on round(position) (merge.py:343-361): the output is
DISCONTINUOUS in the flow where a tile's position (h + 0.5) / s + flow crosses a rounding boundary — a 1e-5 px flow
difference then swaps a row of taps for the opposite one for every pixel of the tile with that sub-pixel phase (measured:
one tile of one case, 6.9e-2 with own flows, 1.8e-7 with the oracle's flows).

The oracle runs are independent: a fork pool computes them on the host cores while the GPU works through the cases."""
import multiprocessing as mp
import os

import numpy as np
import pytest
import torch

import oracle
from helpers import base_config, fuzz_verdict
from handheld_super_resolution import synthetic as synth
import handheld_super_resolution as hsr

pytestmark = pytest.mark.gpu

CFAS = [((0, 1), (1, 2)), ((2, 1), (1, 0)), ((1, 0), (2, 1)), ((1, 2), (0, 1))]
BATCHES = [(0, 22), (1, 22), (2, 20)]  # (generator seed, cases): the 64 cases
if os.environ.get("HHSR_FUZZ_BATCHES"):  # held-out batches, e.g. "10:22,11:22,12:20" (same assertions on other bursts)
    BATCHES = [tuple(int(v) for v in b.split(":")) for b in os.environ["HHSR_FUZZ_BATCHES"].split(",")]
FLIPPED_PER_BATCH = 2   # flipped block-matching decisions (clusters of tiles) per batch  (measured: 0, 0, 1)
# (the per-case rules and their constants: helpers.fuzz_verdict — pure NumPy, unit-tested on the CPU in test_host_logic.py)


def cases(gen_seed, n):
    rng = np.random.default_rng(gen_seed)
    out = []
    for k in range(n):
        ts = int(rng.choice([16, 16, 32]))
        lo, hi = (576, 760) if ts == 16 else (384, 640)  # (default pyramid factors need >= 36 tiles of 16; ts 32: factors 2)
        c = dict(ts=ts, H=int(rng.integers(lo, hi)) // 2 * 2, W=int(rng.integers(lo, hi + 64)) // 2 * 2,
                 nf=int(rng.integers(2, 5)), scale=[1, 1.5, 2, 2, 3][int(rng.integers(0, 5))], cfa=CFAS[int(rng.integers(0, 4))])
        c["wb"] = (float(rng.uniform(1.0, 2.2)), 1.0, float(rng.uniform(1.0, 2.0))) if rng.random() < 0.7 else (1.0, 1.0, 1.0)
        c["iso"] = bool(rng.random() < 0.2)
        c["rob"] = bool(rng.random() < 0.85)
        c["den"] = bool(c["rob"] and rng.random() < 0.25)
        c["seed"] = int(rng.integers(0, 10000))
        c["max_shift"], c["occ"] = float(rng.uniform(0.5, 3.5)), bool(rng.random() < 0.5)
        c["metric0"] = ["L2", "L1", "L1_ref_effective"][(gen_seed + k) % 3] if ts == 16 else "L2"
        c["id"] = f"{gen_seed}.{k}"
        out.append(c)
    return out


def config(c, **hip):
    cfg = base_config(ts=c["ts"], scale=c["scale"], snr=30.0 if c["ts"] == 16 else 18.0,
                      metrics=(c["metric0"], "L2", "L2", "L2"))
    if c["ts"] == 32:
        cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.exif = {"cfa_pattern": [list(r) for r in c["cfa"]], "iso": 100, "white_balance": list(c["wb"])}
    cfg.robustness.enabled = c["rob"]
    cfg.robustness.save_mask = c["rob"]
    if c["iso"]:
        cfg.merging.kernel = "iso"
    if c["den"]:
        cfg.accumulated_robustness_denoiser.enabled = True
        cfg.accumulated_robustness_denoiser.merge.enabled = True
    if hip:
        cfg.hip = hip
    return cfg


def burst(c):
    return synth.make_burst(c["H"], c["W"], c["nf"], seed=c["seed"], max_shift=c["max_shift"], occluder=c["occ"],
                            cfa=c["cfa"], wb=c["wb"])[:2]


def _oracle_case(c):
    os.environ["OMP_NUM_THREADS"] = "1"
    torch.set_num_threads(1)
    ref, comp = burst(c)
    cap = {}
    want, _ = oracle.main(ref, comp, config(c), capture=cap)
    return ref, comp, want, np.stack(cap["flow"]), (np.stack(cap["r"]) if c["rob"] else None), cap["den"]


def check(c, ref, comp, want, oflow, o_r, den_o, report=None):
    """Returns the number of flipped block-matching decisions (clusters of tiles) of the case: 0 or 1."""
    cfg = config(c)
    cfg.debug = True
    out, dbg = hsr.main(ref, comp, cfg)
    o = out.cpu().numpy()
    cfg_i = config(c, inject_flows=[f for f in oflow])
    cfg_i.debug = True
    out_i, dbg_i = hsr.main(ref, comp, cfg_i)
    oi = out_i.cpu().numpy()
    H, W, ts, scale = c["H"], c["W"], c["ts"], c["scale"]
    tag = f"case {c['id']} ({H}x{W} x{c['nf']} s={scale} ts={ts} {c['metric0']} rob={c['rob']} den={c['den']} occ={c['occ']})"
    v, failed = fuzz_verdict((H, W), ts, scale, o, oi, want, np.stack(dbg["flow"]), oflow,
                             np.stack(dbg["robustness"]) if c["rob"] else None,
                             np.stack(dbg_i["robustness"]) if c["rob"] else None, o_r, den_o)
    nflip, one_cluster, n_ica, nan_mis, dflow, dr, dr_i = (v[k] for k in ("nflip", "one_cluster", "n_ica", "nan_mis", "dflow", "dr", "dr_i"))
    inj_max, n_inj, inj_outside, inj_q = v["inj_max"], v["n_inj"], v["inj_outside"], v["inj_q"]
    n_sens, sens_max, n_rest, rest_max, rest_outside = v["n_sens"], v["sens_max"], v["n_rest"], v["rest_max"], v["rest_outside"]
    if report is not None:
        report.append(f"{tag}: flipped {nflip}{'' if one_cluster else ' (NOT one cluster)'}, ica {n_ica}, nan {nan_mis}, flow {dflow:.1e}, r {dr:.1e} / injected {dr_i:.1e}; injected "
                      f"image max {inj_max:.2e} ({n_inj} > 1e-4, {inj_outside} outside rejecting regions); own flows: flow-"
                      f"sensitive {n_sens} (max {sens_max:.1e}), other {n_rest} (max {rest_max:.1e}, {rest_outside} outside "
                      f"rejecting regions); injected outliers x den max {inj_q:.2e}" + (f"  ASSERTIONS FAILED: {'; '.join(failed)}" if failed else ""))
        return int(nflip > 0)
    assert not failed, f"{tag}: " + "; ".join(failed)
    return int(nflip > 0)


_pool, _jobs = None, {}


@pytest.fixture(scope="module")
def oracle_jobs():
    """All 64 oracle runs are submitted at once to a fork pool (NumPy-only children: they never touch the GPU) and the
    batches consume them as they finish: the sweep takes about as long as the slowest oracle case."""
    all_cases = [c for gs, n in BATCHES for c in cases(gs, n)]
    # every core the container may use (cgroup quota: 16 of the GPU boxes' 256 logical CPUs) minus two for this process:
    # with one worker per case the workers starved the checking thread (the sweep took 290 s instead of ~2 min)
    workers = max(1, min(len(all_cases), oracle.available_cores() - 2))
    pool = mp.get_context("fork").Pool(workers)
    jobs = {c["id"]: pool.apply_async(_oracle_case, (c,)) for c in all_cases}
    yield jobs
    pool.terminate()
    pool.join()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("gen_seed,n", BATCHES)
def test_fuzz_sweep(oracle_jobs, gen_seed, n):
    report = [] if os.environ.get("HHSR_FUZZ_REPORT") else None
    flipped = 0
    for c in cases(gen_seed, n):
        flipped += check(c, *oracle_jobs[c["id"]].get(timeout=1500), report=report)
    if report is not None:
        with open(os.environ["HHSR_FUZZ_REPORT"], "a") as f:
            f.write("\n".join(report) + "\n")
    else:
        assert flipped <= FLIPPED_PER_BATCH, f"batch {gen_seed}: {flipped} flipped decisions"

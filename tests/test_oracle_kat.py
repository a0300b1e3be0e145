"""Analytic known-answer tests of the oracle — CPU only."""
import numpy as np

import oracle
from helpers import base_config
from handheld_super_resolution import synthetic as synth


def _smooth(rng, h, w, sigma=1.5):
    from scipy.ndimage import gaussian_filter

    f = gaussian_filter(rng.standard_normal((h, w)), sigma)
    return ((f - f.min()) / (f.max() - f.min())).astype(np.float32)


def test_block_matching_recovers_planted_integer_shift():
    rng = np.random.default_rng(0)
    big = _smooth(rng, 160, 160)
    ref = big[16:144, 16:144]
    for metric in ("l2", "l1"):
        for (sy, sx) in ((3, -2), (0, 4), (-4, -4)):
            mov = big[16 + sy:144 + sy, 16 + sx:144 + sx]
            flow0 = np.zeros((8, 8, 2), np.float32)
            f = oracle.bm_l2(ref, mov, flow0, 16, 4) if metric == "l2" else oracle.bm_l1(ref, mov, flow0, 16, 4)
            inner = f[1:-1, 1:-1]
            # mov(p) = scene(p + s) so moving(p - s) = ref(p): flow = -s
            assert (inner[..., 0] == -sx).all() and (inner[..., 1] == -sy).all(), (metric, sy, sx)


def test_ica_takes_half_steps():
    """D9: with the un-normalised gradient filter each iteration halves the residual."""
    x = np.arange(96, dtype=np.float64)
    X, Y = np.meshgrid(x, x)
    img = lambda dx, dy: (0.5 + 0.2 * np.sin(2 * np.pi * (X + dx) / 23.0) + 0.2 * np.cos(2 * np.pi * (Y + dy) / 31.0)).astype(np.float32)
    ref, mov = img(0, 0), img(-0.4, 0.3)  # moving(p + (0.4, -0.3)) = ref(p)
    gx, gy, H = oracle.init_ica(ref, 32)
    res = []
    for n in (1, 2, 3):
        f = oracle.ica(ref, gx, gy, H, mov, np.zeros((3, 3, 2), np.float32), 32, n)
        res.append(np.abs(f[1, 1] - np.array([0.4, -0.3])).max())
    assert 0.15 < res[0] < 0.25 and res[1] < 0.6 * res[0] and res[2] < 0.6 * res[1]


def test_identical_frames_zero_flow_full_robustness_and_border():
    ref, _, _ = synth.make_burst(512, 512, 1, seed=3)
    cfg = base_config(ts=16, scale=1)
    cap = {}
    oracle.main(ref, ref[None], cfg, capture=cap)
    assert np.abs(cap["flow"][0]).max() == 0
    r = cap["r"][0]
    assert (r[:3] == 0).all() and (r[:, :3] == 0).all()  # D6
    assert (r[3:, 3:] == 1).all()  # clamp(s2 * exp(0) - t, 0, 1)


def test_constant_colour_is_reproduced():
    H = W = 512
    cfa = np.array([[0, 1], [1, 2]])
    rgb = (0.3, 0.5, 0.2)
    raw = np.empty((H, W), np.float32)
    for i in range(2):
        for j in range(2):
            raw[i::2, j::2] = rgb[cfa[i, j]]
    cfg = base_config(ts=16, scale=2)
    out, _ = oracle.main(raw, raw[None], cfg)
    inner = out[8:-8, 8:-8]
    for c in range(3):
        assert np.abs(inner[..., c] - rgb[c]).max() < 1e-5


def test_hr_pixels_missing_a_colour_are_nan():
    """D6/D10 consequence: with r = 0 on the border only the ref frame contributes there, and an HR pixel
    whose 3x3 ref window holds no sample of some colour ends 0/0 = NaN (kept; the CLI nan_to_num's it)."""
    ref, comp, _ = synth.make_burst(512, 512, 2, seed=4)
    cfg = base_config(ts=16, scale=2)
    out, _ = oracle.main(ref, comp, cfg)
    assert np.isnan(out[:, -1, 0]).any() or np.isnan(out[-1, :, 0]).any()
    assert np.isfinite(out[16:-16, 16:-16]).all()


def test_parallel_oracle_is_bit_identical_to_sequential():
    """oracle.main_parallel (one worker process per frame, contributions added in frame order — the all-cores CPU
    baseline of bench.py) == oracle.main, bit for bit, including the accumulated robustness."""
    ref, comp, _ = synth.make_burst(128, 160, 4, seed=11, max_shift=2.0, occluder=True)
    def cfg0():
        c = base_config(ts=16, scale=2)
        c.block_matching.tuning.factors = [1, 2, 2, 2]
        c.robustness.save_mask = True
        return c
    want, wdbg = oracle.main(ref, comp, cfg0())
    cap = {}
    got, gdbg, used = oracle.main_parallel(ref, comp, cfg0(), workers=3, capture=cap)
    assert used == 3
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(gdbg["accumulated robustness"], wdbg["accumulated robustness"])
    assert len(cap["flow"]) == 3 and all(f is not None for f in cap["flow"])
    got1, _, used1 = oracle.main_parallel(ref, comp, cfg0(), workers=1)
    assert used1 == 1 and np.array_equal(got1, want, equal_nan=True)


def _ulp_report(a, b):
    """(fraction bit-identical incl. NaN == NaN, largest difference in float32 ulps of the larger operand)."""
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(all="ignore"):
        ulp = np.where(same, 0.0, np.abs(a.astype(np.float64) - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)))
    return float(same.mean()), float(np.nanmax(ulp)) if ulp.size else 0.0


def test_c_merge_equals_numpy_merge():
    """oracle.cfast (oracle/csrc/merge.c: the accumulation compiled with gcc, what the sweeps and full-size comparisons
    run) against oracle/merge.py (the form pinned by the reference's own outputs): num, den of Alg. 4 and Alg. 11 on every
    scale family, iso kernels, grey mode, the accumulated-robustness rules (widened + overwriting reference splat), NaN
    covariances (flat regions, D10), negative covariance fractions at the border (D11) and flows that push windows and
    whole tiles over the image border.  Bit-identical, or one float32 ulp where libm's exp and NumPy's differ."""
    from oracle import cfast
    import importlib

    npm = importlib.import_module("oracle.merge")
    rng = np.random.default_rng(7)
    worst = 0.0
    for (H, W, scale, kernel, mode, den) in ((64, 80, 2, "steerable", "bayer", False), (64, 80, 1.5, "steerable", "bayer", True),
                                              (48, 64, 3, "steerable", "bayer", False), (64, 64, 1, "iso", "bayer", True),
                                              (48, 48, 2, "steerable", "grey", False), (48, 64, 2, "iso", "grey", False)):
        cfg = base_config(ts=16, scale=scale)
        cfg.merging.kernel = kernel
        cfg.mode = mode
        cfg.exif.cfa_pattern = [[1, 2], [0, 1]]
        if den:
            cfg.accumulated_robustness_denoiser.enabled = True
        comp = rng.uniform(0, 1, (H, W)).astype(np.float32)
        flow = rng.uniform(-3, 3, (H // 16, W // 16, 2)).astype(np.float32)
        flow[0, 0] = (-20.5, 7.25)  # a tile pushed over the left border
        flow[-1, -1] = (30.0, 30.0)  # ... and out of the frame
        ch, cw = (H // 2, W // 2) if mode == "bayer" else (H, W)
        a = rng.uniform(0.3, 2.0, (ch, cw)).astype(np.float32)
        b = rng.uniform(-0.25, 0.25, (ch, cw)).astype(np.float32)
        covs = np.stack([np.stack([a, b], -1), np.stack([b, a * 0.7], -1)], -2).astype(np.float32)
        covs[3:6, 4:9] = np.nan  # D10
        covs[10, 10] = 0         # singular
        r = rng.uniform(0, 1, (H, W)).astype(np.float32)
        r[:3] = 0
        osz = (round(scale * H), round(scale * W))
        base = rng.uniform(0, 2, (*osz, 3)).astype(np.float32)
        acc_rob = rng.uniform(0, 12, (H, W)) if den else None
        outs = []
        for m in (npm, cfast):
            num, dn = base.copy(), base.copy() * 0.5
            m.merge(comp, flow, covs, r, num, dn, np.array(cfg.exif.cfa_pattern), cfg)
            mid = (num.copy(), dn.copy())
            m.merge_ref(comp[::-1].copy(), covs, num, dn, np.array(cfg.exif.cfa_pattern), cfg, acc_rob)
            outs.append(mid + (num, dn))
        for x, y in zip(*outs):
            frac, ulps = _ulp_report(x, y)
            worst = max(worst, ulps)
            assert frac > 0.9999 and ulps <= 1.0, (H, W, scale, kernel, mode, den, frac, ulps)
    # the whole pipeline through it (and main_parallel), incl. flows= / reuse= (the two-sided flow injection)
    ref, comp, _ = synth.make_burst(128, 160, 3, seed=11, max_shift=2.0, occluder=True)
    def cfg0():
        c = base_config(ts=16, scale=2)
        c.block_matching.tuning.factors = [1, 2, 2, 2]
        c.robustness.save_mask = True
        return c
    cap = {}
    want, _ = oracle.main(ref, comp, cfg0(), capture=cap)
    got, _ = oracle.main(ref, comp, cfg0(), fast=True)
    frac, ulps = _ulp_report(got, want)
    assert frac > 0.9999 and ulps <= 2.0, (frac, ulps)
    cap2 = {}
    again, _ = oracle.main(ref, comp, cfg0(), fast=True, flows=cap["flow"], reuse=cap, capture=cap2)
    assert np.array_equal(again, got, equal_nan=True) and np.array_equal(cap2["r"][1], cap["r"][1])
    par, _, _ = oracle.main_parallel(ref, comp, cfg0(), workers=2, fast=True, flows=cap["flow"])
    assert np.array_equal(par, got, equal_nan=True)
    shifted = [f + np.float32(0.25) for f in cap["flow"]]
    other, _ = oracle.main(ref, comp, cfg0(), fast=True, flows=shifted, reuse=cap)
    assert not np.array_equal(other, got, equal_nan=True)

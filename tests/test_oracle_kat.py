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

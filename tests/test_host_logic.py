"""Host-side logic (config shim, parameter derivation, level shapes, filter taps) — CPU only."""
import numpy as np
import pytest

import oracle
from helpers import base_config
import handheld_super_resolution as hsr
from handheld_super_resolution import params, utils_image
from handheld_super_resolution.config import Config, OmegaConf, default_config


def test_config_shim():
    cfg = default_config()
    assert cfg.block_matching.tuning.factors == [1, 2, 4, 4]
    assert cfg.noise_model.get("alpha", None) is None
    cfg.noise_model.update({"alpha": 1.0, "beta": 2.0})
    assert cfg.noise_model.alpha == 1.0 and cfg["noise_model"]["beta"] == 2.0
    cfg.exif = {"cfa_pattern": [[0, 1], [1, 2]]}
    assert isinstance(cfg.exif, Config) and cfg.exif.cfa_pattern[1][1] == 2
    with pytest.raises(AttributeError):
        cfg.nope
    merged = OmegaConf.merge(cfg, OmegaConf.from_dotlist(["scale=2", "robustness.tuning.t=0.2", "merging.kernel=iso"]))
    assert merged.scale == 2 and merged.robustness.tuning.t == 0.2 and merged.merging.kernel == "iso"
    assert merged.robustness.tuning.s1 == 2  # untouched siblings survive the merge
    c2 = cfg.copy()
    c2.scale = 3
    assert cfg.scale == 1


@pytest.mark.parametrize("snr", [3.0, 6.0, 10.0, 14.0, 14.5, 22.0, 22.5, 27.3, 30.0, 45.0])
def test_params_match_oracle_and_reference(golden, snr):
    a, b = default_config(), default_config()
    params.update_snr_config(a, snr)
    oracle.update_snr_config(b, snr)
    assert OmegaConf.to_container(a) == OmegaConf.to_container(b)
    row = [r for r in golden("params")["table"] if r[0] == snr][0]
    t = a.merging.tuning
    np.testing.assert_allclose([a.block_matching.tuning.tile_size, *a.block_matching.tuning.tile_sizes, t.k_detail,
                                t.k_denoise, t.D_th, t.D_tr], row[1:], rtol=0, atol=1e-15)


def test_level_shapes_table():
    """SURVEY.md App. D."""
    cases = {((3000, 4000), 16): ([(3008, 4000), (1500, 1996), (371, 495), (88, 119)],
                                  [(3000, 4000), (1496, 1996), (370, 495), (88, 119)],
                                  [(188, 250), (93, 124), (23, 30), (11, 14)]),
             ((3000, 4000), 64): ([(3008, 4032), (1500, 2012), (371, 499), (88, 120)], None,
                                  [(47, 63), (23, 31), (5, 7), (2, 3)]),
             ((512, 512), 16): ([(512, 512), (252, 252), (59, 59), (10, 10)], None, [(32, 32), (15, 15), (3, 3), (1, 1)])}
    for (shape, ts), (ref, mov, tiles) in cases.items():
        cfg = base_config(ts=ts)
        r, m, t = params.level_shapes(shape, cfg)
        assert r == ref and t == tiles
        if mov:
            assert m == mov
        assert (r, m, t) == oracle.level_shapes(shape, cfg)


def test_sanitize():
    cfg = base_config(ts=16)
    params.sanitize_config(cfg, (512, 512))
    with pytest.raises(ValueError):  # D8: 512x512 with Ts=32 has no tile at the coarsest level
        params.sanitize_config(base_config(ts=32), (512, 512))
    bad = base_config(ts=16)
    bad.robustness.enabled = False
    with pytest.raises(ValueError):
        params.sanitize_config(bad, (512, 512))  # save_mask without robustness
    bad = base_config(ts=16)
    bad.block_matching.tuning.flow_upscale_mode = "cubic"
    with pytest.raises(AssertionError):
        params.sanitize_config(bad, (512, 512))


def test_gaussian_taps_match_scipy():
    from scipy.ndimage._filters import _gaussian_kernel1d

    for f in (2, 4):
        want = _gaussian_kernel1d(sigma=f * 0.5, order=0, radius=int(4 * f * 0.5 + 0.5))[::-1].astype(np.float32)
        np.testing.assert_array_equal(utils_image.gaussian_taps(f), want)
        np.testing.assert_array_equal(oracle.gaussian_taps(f)[0], want)


def test_prepare_config_derives_like_process():
    from handheld_super_resolution import synthetic as synth

    ref, _, _ = synth.make_burst(160, 160, 1, seed=1)
    cfg = default_config()
    cfg.verbose = 0
    cfg.block_matching.tuning.tile_size = 16
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    hsr.prepare_config(cfg, ref, synth.ALPHA_ISO100, synth.BETA_ISO100, [[0, 1], [1, 2]], [2.0, 1.0, 1.5, 1.0])
    assert cfg.block_matching.tuning.tile_sizes == [16, 16, 16, 8]
    assert cfg.merging.tuning.k_detail == 0.25 and cfg.merging.tuning.D_tr == 1.0  # SNR clipped to 30
    assert cfg.accumulated_robustness_denoiser.enabled is False
    assert len(cfg.noise_model.std_curve) == 1001 and cfg.exif.white_balance[0] == 2.0


# ------------------------------------------------------------------------------------------ burst front end
def test_oracle_normalize_known_answers():
    """utils_dng.py:149-160 on a 2x2 CFA cell with hand-computed float32 results."""
    raw = np.array([[[1088, 4159], [64, 2112]]], dtype=np.uint16)
    cfa = [[0, 1], [1, 2]]
    out = oracle.frontend.normalize_burst(raw, [64, 64, 64], 4159, [2.0, 1.0, 1.5], cfa)
    f = np.float32
    want = np.array([[[(f(1088) - f(64)) / f(4095) * f(2.0), f(1.0)], [f(0.0), (f(2112) - f(64)) / f(4095) * f(1.5)]]])
    assert out.dtype == np.float32 and np.array_equal(out, want.astype(np.float32))


def test_monte_carlo_noise_curves_cpu():
    """Seeded Monte-Carlo estimator (torch, here on the CPU): reproducible, equals the analytic un-clipped limits
    where no clipping occurs, agrees with the NumPy restatement of the reference's unitary_MC where it does."""
    from handheld_super_resolution import fast_monte_carlo as mc, synthetic as synth

    a, b = synth.ALPHA_ISO100 * 8, synth.BETA_ISO100 * 8
    s1, d1 = mc.run_fast_MC(a, b, seed=3, device="cpu", n_patches=20000)
    s2, d2 = mc.run_fast_MC(a, b, seed=3, device="cpu", n_patches=20000)
    assert s1.shape == (1001,) and np.array_equal(s1, s2) and np.array_equal(d1, d2)
    sa, da = synth.noise_curves(a, b)
    xmin, xmax = mc.get_non_linearity_bound(a, b)
    assert (xmin, xmax) == oracle.frontend.non_linearity_bound(a, b)
    lo, hi = int(np.ceil(xmin * 1000)) + 2, int(np.floor(xmax * 1000)) - 2
    assert np.abs(s1[lo:hi] / sa[lo:hi] - 1).max() < 0.01 and np.abs(d1[lo:hi] / da[lo:hi] - 1).max() < 0.02
    assert s1[0] < 0.75 * sa[0] and s1[1000] < 0.75 * sa[1000]  # clipping at 0 / 1 shrinks the spread
    rng = np.random.default_rng(0)
    for i in (0, 3, 998, 1000):
        dm, sm = oracle.frontend.unitary_mc(a, b, i / 1000, 20000, rng)
        assert abs(s1[i] / sm - 1) < 0.03 and abs(d1[i] / dm - 1) < 0.05


# ------------------------------------------------------------------------------------------ graph-replay guards
def test_config_watch_sees_every_in_place_edit():
    """a sum of per-node edit counters can cancel out (a nested mapping replaced by a fresh one), and
    dict.setdefault / pop / clear and in-place list edits bypass __setitem__.  ConfigWatch compares (identity, counter)
    per node plus the content of short lists, and Config routes every mutating method through its counter."""
    from handheld_super_resolution.graph import ConfigWatch
    from handheld_super_resolution.config import Config

    def fresh():
        cfg = base_config()
        w = ConfigWatch()
        assert not w.changed(cfg) and not w.changed(cfg)  # first look never counts as a change
        return cfg, w

    cfg, w = fresh()
    cfg.scale = 3
    assert w.changed(cfg) and not w.changed(cfg)
    cfg, w = fresh()  # nested mapping replaced by a fresh one with FEWER edits: a counter sum would not move
    cfg.robustness.tuning.t = 0.2
    cfg.robustness.tuning.t = 0.3
    assert w.changed(cfg)
    before = sum(n.version() for n in (cfg, cfg.robustness, cfg.robustness.tuning))
    cfg.robustness["tuning"] = Config({"t": 0.5, "s1": 2, "s2": 12, "Mt": 0.8})
    object.__setattr__(cfg.robustness, "_ver", cfg.robustness.version() - 1)  # even with the parent's counter rolled back
    assert w.changed(cfg), before
    cfg, w = fresh()
    cfg.block_matching.tuning.tile_sizes[0] = 8  # in-place list edit
    assert w.changed(cfg)
    for edit in (lambda c: c.setdefault("new_key", 1), lambda c: c.pop("grey_method"), lambda c: c.robustness.clear(),
                 lambda c: c.merging.popitem()):
        cfg, w = fresh()
        edit(cfg)
        assert w.changed(cfg)
    a, b = base_config(), base_config()
    w = ConfigWatch()
    w.changed(a)
    assert not w.changed(b)  # another object: a first look again


def test_fuzz_verdict_rules():
    """The per-case rules of the randomised parity sweep (helpers.fuzz_verdict, used by tests/test_fuzz_parity.py on the GPU)
    on hand-made arrays: what passes, and that each rule trips on the violation it exists for.  The contract compares on
    identical inputs: o (HIP, own flows) against want_h (oracle on HIP's flows) and want_hm (oracle's merge on HIP's flows
    and HIP's robustness), oi (HIP on the oracle's flows) against want / want_om."""
    from helpers import fuzz_verdict

    H, W, ts, scale, n = 64, 96, 16, 2, 2
    ny, nx = H // ts, W // ts
    rng = np.random.default_rng(0)
    want = rng.random((2 * H, 2 * W, 3)).astype(np.float32)
    want[0, 0, 1] = np.nan
    want[1, 1, 2] = np.inf
    oflow = rng.standard_normal((n, ny, nx, 2)).astype(np.float32)
    o_r = np.ones((n, H, W), np.float32)
    o_r[0, 20:30, 40:50] = 0.3          # a region where frame 0 is being rejected
    den = np.full_like(want, 2.0)

    def run(o=None, oi=None, want_h=None, want_hm=None, want_om=None, gflow=None, hr=None, hr_i=None, o_r_h=None, den_o=None,
            den_h=None, rob=True):
        wh = want.copy() if want_h is None else want_h
        o = wh.copy() if o is None else o
        oi = want.copy() if oi is None else oi
        whm = (wh if want_hm is None else want_hm) if rob else None
        wom = (want if want_om is None else want_om) if rob else None
        dh, do = den if den_h is None else den_h, den if den_o is None else den_o
        return fuzz_verdict((H, W), ts, scale, o, oi, want, wh, oflow if gflow is None else gflow, oflow,
                            (o_r if hr is None else hr) if rob else None, (o_r if hr_i is None else hr_i) if rob else None,
                            o_r if rob else None, (o_r if o_r_h is None else o_r_h) if rob else None, do, dh,
                            whm, dh if rob else None, wom, do if rob else None)

    v, failed = run()
    assert not failed and v["side_h"]["n"] == 0 and v["side_o"]["n"] == 0 and v["nflip"] == 0  # NaN == NaN, inf == inf
    assert v["side_h"]["nan_mis"] == 0 and v["side_h"]["m_nan"] == 0 and not run(rob=False)[1]
    # arithmetic noise below 1e-4 everywhere
    assert not run(o=want + 5e-5, oi=want - 5e-5)[1]
    # a value above 1e-4 where every frame is accepted fails on EITHER side, whatever the merge comparison says
    bad = want.copy()
    bad[100, 20, 0] += 3e-4
    assert run(oi=bad)[1] and run(o=bad)[1] and run(o=bad, want_hm=bad)[1]
    # ... inside the rejecting region (HR rows 40-60, cols 80-100) it needs an EXPLANATION: the oracle's merge on the same
    # flows and HIP's robustness must reproduce it (then it is the effect of the <= 1e-4 by which r differs) — any size
    ok = want.copy()
    ok[50, 90, 0] += 0.05
    ok[44:48, 84:88, :] += 2e-3          # and any number of them
    for side, kw in (("side_o", dict(oi=ok, want_om=ok)), ("side_h", dict(o=ok, want_hm=ok))):
        v, failed = run(**kw)
        assert not failed and v[side]["n"] == 49 and v[side]["outside"] == 0 and v[side]["unexplained"] == 0 and v[side]["m_n"] == 0
    v, failed = run(o=ok)                # not reproduced: fails — and shows up in the merge comparison
    assert failed and v["side_h"]["unexplained"] == 49 and v["side_h"]["m_n"] == 49
    assert run(oi=ok)[1]
    # merge alone: nothing above 1e-4, also where the accumulated weight vanishes (round 4's numerator rule is gone)
    tiny = den.copy()
    tiny[50, 90, 0] = 1e-6
    one = want.copy()
    one[50, 90, 0] += 0.05
    assert run(o=one)[1] and run(o=one, den_h=tiny)[1] and run(oi=one, den_o=tiny)[1]
    # the rejecting region of a side is where EITHER robustness map of that side rejects
    r_h = np.ones_like(o_r)
    assert not run(o_r_h=r_h, hr=r_h)[1]
    assert run(o=ok, want_hm=ok, o_r_h=r_h, hr=r_h)[1]   # nothing is being rejected under HIP's flows there: no excuse
    r_a, r_b = np.ones_like(o_r), np.ones_like(o_r)
    r_a[0, 20:30, 40:50], r_b[0, 20:30, 40:50] = 0.99895, 0.99902   # 7e-5 apart, on either side of the 0.999 mark
    assert not run(o=ok, want_hm=ok, o_r_h=r_b, hr=r_a)[1] and not run(o=ok, want_hm=ok, o_r_h=r_a, hr=r_b)[1]
    # a flow-sensitive jump is fine when — and only when — the oracle reproduces it on HIP's flows (no magnitude cap)
    jump = want.copy()
    jump[100:110, 20:30, :] += 0.6
    v, failed = run(o=jump, want_h=jump)
    assert not failed and v["n_own"] == 300 and v["n_orc"] == 300 and abs(v["orc_max"] - 0.6) < 1e-6
    assert run(o=jump)[1]                # HIP alone jumps: fails, whatever the size
    # flows: one flipped 2 x 2 block is one decision; what it does to the image must be in want_h too
    g = oflow.copy()
    g[1, 1:3, 2:4] += 0.08
    o2 = want.copy()
    o2[2 * ts * 1: 2 * ts * 3, 2 * ts * 2: 2 * ts * 4] += 0.05
    v, failed = run(gflow=g, o=o2, want_h=o2)
    assert not failed and v["nflip"] == 4 and v["one_cluster"] and v["n_own"] == 0  # (inside the footprint: not counted)
    assert run(gflow=g, o=o2)[1]         # the oracle on HIP's flows does not show it: fails
    g[0, 0, 0] += 0.08                   # a second decision in another frame
    assert run(gflow=g)[1]
    g = oflow.copy()
    g[1, 0:4, 0:6] += 0.08               # one cluster, but 24 tiles (MAX_FLIP_TILES = 16)
    assert run(gflow=g)[1]
    g = oflow.copy()
    g[0, 2, 3] += 3e-4                   # an ill-conditioned ICA tile: counted, tolerated
    v, failed = run(gflow=g)
    assert not failed and v["n_ica"] == 1 and v["nflip"] == 0
    g[0] += 3e-4                         # ... not a whole frame of them
    assert run(gflow=g)[1]
    g = oflow.copy()
    g[0, 2, 3] += 9e-5                   # below 1e-4 px: flow agreement
    assert not run(gflow=g)[1] and run(gflow=g)[0]["dflow"] > 8e-5
    # robustness beyond 1e-4 ANYWHERE, on either side; NaN pattern (no footprint exemption); inf vs finite
    hr = o_r.copy()
    hr[1, 5, 5] -= 2e-4
    assert run(hr=hr)[1] and run(hr_i=hr)[1]
    nanned = want.copy()
    nanned[3, 3, 0] = np.nan
    assert run(o=nanned)[1] and run(oi=nanned)[1] and not run(o=nanned, want_h=nanned)[1]
    assert run(o=nanned, want_h=nanned, want_hm=want)[1]  # (the merge comparison has its own NaN pattern rule)
    finite = want.copy()
    finite[1, 1, 2] = 1.0
    assert run(oi=finite)[1]
    # HIP's accumulated robustness must be the sum of its own maps (float32 accumulation noise only)
    from helpers import side_failures, ACC_TOL
    ok_side = dict(run()[0]["side_h"], acc=3e-7)
    assert not side_failures("x", ok_side) and side_failures("x", dict(ok_side, acc=2 * ACC_TOL))


def test_robustness_sum_keeps_float64_decisions():
    """robustness.RobustnessSum: the accumulated-robustness denoiser's `<=` / `<` comparisons with max_frame_count are those
    of the reference's float64 sum, although the kernel reads a float32 map (found by the round-5 sweep, case 4300.15)."""
    import torch
    from handheld_super_resolution.robustness import RobustnessSum

    one_m = np.float32(1.0) - np.float32(2.0 ** -24)          # 0.99999994
    r = [torch.tensor([[1.0, 1.0, 1.0, 0.5]], dtype=torch.float32), torch.tensor([[1.0, 1.0, 1.0, 0.5]], dtype=torch.float32),
         torch.tensor([[float(one_m), 1.0, 0.25, 1.0]], dtype=torch.float32)]
    acc = RobustnessSum((1, 4), "cpu")
    f32 = torch.zeros((1, 4), dtype=torch.float32)
    for t in r:
        acc.add(t)
        f32 += t
    assert float(f32[0, 0]) == 3.0 and float(acc.sum[0, 0]) < 3.0    # the float32 sum has lost the decision
    for mfc in (3, 3.0, 2.25, 8, 2.0000001):
        a = acc.for_decisions(mfc).to(torch.float64)
        assert torch.equal(a <= mfc, acc.sum <= mfc) and torch.equal(a < mfc, acc.sum < mfc), mfc
    a = acc.for_decisions(3)
    assert float(a[0, 0]) < 3.0 and float(a[0, 1]) == 3.0 and a.dtype == torch.float32
    assert torch.equal(acc.mask(), acc.sum.to(torch.float32)) and acc.mask((0, 1)).shape == (1, 4)
    # randomised: float64 sums of up to 20 float32 maps, dense around every threshold (also thresholds float32 cannot hold)
    rng = np.random.default_rng(5)
    for mfc in (1, 2, 3, 8, 19, 2.5, 2.1, 0.1, 16777217.0 / 8388608.0):
        m32 = np.float32(mfc)
        near = np.concatenate([np.float64(m32) + np.arange(-40, 41) * 2.0 ** -27,            # +- 5 float32 ulps of 2, float64 steps
                               np.float64(mfc) + np.arange(-3, 4) * np.spacing(np.float64(mfc)),
                               rng.uniform(0, 20, 256), [0.0, np.float64(mfc), np.float64(m32), 20.0]])
        if float(m32) != float(mfc):  # no float32 EQUALS such a threshold: `sum == mfc` itself cannot be handed to the kernel
            near = near[near != np.float64(mfc)]
        s64 = torch.from_numpy(np.maximum(near, 0.0)[None])
        a = RobustnessSum.decisions_of(s64, mfc)
        assert a.dtype == torch.float32
        a64 = a.to(torch.float64)
        assert torch.equal(a64 <= mfc, s64 <= mfc) and torch.equal(a64 < mfc, s64 < mfc), mfc
        far = (s64 - float(mfc)).abs() > 4 * float(np.spacing(m32))  # away from the threshold the map IS the rounded sum
        assert torch.equal(a[far], s64.to(torch.float32)[far])

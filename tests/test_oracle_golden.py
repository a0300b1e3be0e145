"""Pin the CPU oracle against vectors captured from the reference's own code (tests/golden/*.npz,
produced by tools/refsim/make_goldens.py in the build container).  CPU only."""
import numpy as np
import pytest

import oracle
from helpers import base_config, acc_pattern, assert_close

F32 = np.float32


def test_params(golden):
    from handheld_super_resolution.config import default_config

    for row in golden("params")["table"]:
        cfg = default_config()
        oracle.update_snr_config(cfg, row[0])
        t = cfg.merging.tuning
        got = [cfg.block_matching.tuning.tile_size, *cfg.block_matching.tuning.tile_sizes,
               t.k_detail, t.k_denoise, t.D_th, t.D_tr]
        np.testing.assert_allclose(got, row[1:], rtol=0, atol=1e-15)


def test_grey(golden):
    g = golden("grey")
    for tag in "abc":
        assert_close(oracle.grey_fft(g["in_" + tag]), g["out_" + tag], 0, 2e-6, "grey " + tag)
    assert_close(oracle.decimate_to_grey(g["dec_in"]), g["dec_out"], 0, 1e-7, "decimate")


def test_downsample(golden):
    g = golden("downsample")
    assert_close(oracle.downsample(g["img"], 2), g["f2"], 0, 1e-6, "f2")
    assert_close(oracle.downsample(g["img"], 4), g["f4"], 0, 1e-6, "f4")
    pyr = oracle.build_gaussian_pyramid(g["img2"], [1, 2, 4, 2])
    for i in range(4):
        assert_close(pyr[3 - i], g[f"pyr{i}"], 0, 1e-6, f"pyr{i}")


def test_hessian(golden):
    g = golden("hessian")
    for ts in (8, 16, 32, 64):
        gx, gy, H = oracle.init_ica(g["lvl"], ts)
        assert_close(gx, g["gx"], 0, 0, "gx")
        assert_close(gy, g["gy"], 0, 0, "gy")
        assert_close(H, g[f"H{ts}"], 1e-5, 1e-7, f"H{ts}")


def test_bm_l2(golden):
    g = golden("bm_l2")
    for tag in ("t16", "t8", "t32"):
        ts, r = (int(v) for v in g[tag + "_ts_r"])
        flow, cost = oracle.bm_l2(g[tag + "_ref"], g[tag + "_mov"], g[tag + "_flow_in"], ts, r, return_cost=True)
        want = g[tag + "_flow_out"]
        diff = np.abs(flow - want).max(-1) > 0
        # any disagreement must be a near-tie of the two best costs (FFT float32 vs exact sums)
        for ty, tx in zip(*np.nonzero(diff)):
            c = np.sort(cost[ty, tx].ravel())
            assert (c[1] - c[0]) <= 1e-4 * max(1.0, abs(c[0])), (tag, ty, tx, c[:3])
        assert diff.mean() <= 0.05, (tag, diff.mean())


def test_ica(golden):
    g = golden("ica")
    for ts in (8, 16, 32, 64):
        ref, mov = g[f"t{ts}_ref"], g[f"t{ts}_mov"]
        gx, gy, H = oracle.init_ica(ref, ts)
        assert_close(H, g[f"t{ts}_H"], 1e-5, 1e-7, f"H ts={ts}")
        flow = oracle.ica(ref, gx, gy, g[f"t{ts}_H"], mov, g[f"t{ts}_flow_in"], ts, 3)
        assert_close(flow, g[f"t{ts}_flow_out"], 1e-4, 2e-5, f"ica ts={ts}")


def test_upscale(golden):
    g = golden("upscale")
    for mode in ("nearest", "bilinear", "bicubic"):
        cfg = base_config(ts=16)
        cfg.block_matching.tuning.flow_upscale_mode = mode
        assert_close(oracle.upscale_lvl(g["flow"], (11, 15), 2, cfg), g[mode + "_l2"], 1e-6, 1e-6, mode + " l2")
        assert_close(oracle.upscale_lvl(g["flow"], (21, 29), 1, cfg), g[mode + "_l1"], 1e-6, 1e-6, mode + " l1")


def test_kernels(golden):
    g = golden("kernels")
    for law in ("linear", "hard_threshold"):
        cfg = base_config()
        cfg.merging.selection_law = law
        covs = oracle.estimate_kernels(g["raw"], cfg)
        if law == "linear":
            assert np.isnan(g["cov_" + law][:3, :3]).all()  # the constant block (D10)
        else:
            assert np.isfinite(g["cov_" + law]).all()  # NaN anisotropy falls into the k1 = k2 = 1 branch
        assert_close(covs, g["cov_" + law], 2e-5, 1e-7, "cov " + law)
    cfg = base_config()
    cfg.merging.tuning.update({"k_detail": "SNR_based", "k_denoise": "SNR_based", "D_th": "SNR_based", "D_tr": "SNR_based"})
    oracle.update_snr_config(cfg, 10.0)
    assert_close(oracle.estimate_kernels(g["raw"], cfg), g["cov_snr10"], 2e-5, 1e-7, "cov snr10")


def test_robustness(golden):
    g = golden("robustness")
    cfa, wb = g["cfa"], g["wb"]
    guide = oracle.guide_image(g["ref"], cfa, wb)
    assert_close(guide, g["guide"], 1e-7, 0, "guide")
    m, v = oracle.local_stats(guide)
    assert_close(m, g["gmeans"], 1e-6, 1e-8, "means")
    assert_close(v, g["gvars"], 1e-5, 1e-9, "vars")
    cfg = base_config(ts=16)
    rm, rv = oracle.init_robustness(g["ref"], cfa, wb, cfg)
    assert np.isinf(g["ref_means"][:, 0, :]).all() and np.isinf(g["ref_means"][:, :, 0]).all()  # D6
    assert_close(rm, g["ref_means"], 1e-6, 1e-8, "ref means")
    assert_close(rv, g["ref_vars"], 1e-5, 1e-9, "ref vars")
    dbg = {}
    r = oracle.compute_robustness(g["comp"], g["ref_means"], g["ref_vars"], g["flow"], cfa, wb,
                                  (g["std_curve"], g["diff_curve"]), cfg, debug=dbg)
    assert_close(dbg["comp_means_up"], g["comp_means_up"], 1e-6, 1e-8, "warped means")
    assert_close(dbg["S"], g["S"], 0, 0, "S")
    assert_close(dbg["sigma_sq"], g["sigma_sq"], 1e-5, 0, "sigma_sq")
    assert_close(dbg["d_sq"], g["d_sq"], 1e-4, 1e-12, "d_sq")
    assert_close(dbg["R"], g["R"], 1e-4, 1e-5, "R")
    assert_close(r, g["r"], 1e-4, 1e-5, "r")
    assert (g["r"][:3] == 0).all() and (g["r"][:, :3] == 0).all()  # D6: first 3 rows / columns


def _merge_form(form):
    import importlib

    return importlib.import_module("oracle.merge" if form == "numpy" else "oracle.cfast")


@pytest.mark.parametrize("tag,scale,kern,do_ref", [("s2", 2, "steerable", True), ("s15", 1.5, "steerable", True),
                                                   ("s1", 1, "steerable", True), ("s3", 3, "steerable", False),
                                                   ("s2iso", 2, "iso", True)])
@pytest.mark.parametrize("form", ["numpy", "c"])
def test_merge(golden, tag, scale, kern, do_ref, form):
    """Both forms of the accumulation — oracle/merge.py and its C restatement oracle/csrc/merge.c (oracle.cfast) —
    against the outputs of the reference's own accumulate / accumulate_ref."""
    g = golden("merge")
    om = _merge_form(form)
    H, W = g["comp"].shape
    cfg = base_config(ts=16, scale=scale)
    cfg.merging.kernel = kern
    oh, ow = round(scale * H), round(scale * W)
    num, den = acc_pattern(oh, ow, 0), acc_pattern(oh, ow, 5)
    om.merge(g["comp"], g["flow"], g["covs"], g["r"], num, den, g["cfa"], cfg)
    assert_close(num, g[tag + "_num"], 2e-6, 1e-7, tag + " num")
    assert_close(den, g[tag + "_den"], 2e-6, 1e-7, tag + " den")
    if do_ref:
        num, den = acc_pattern(oh, ow, 0), acc_pattern(oh, ow, 5)
        om.merge_ref(g["ref"], g["covs_ref"], num, den, g["cfa"], cfg)
        assert_close(num, g[tag + "_numref"], 1e-5, 1e-7, tag + " numref")
        assert_close(den, g[tag + "_denref"], 1e-5, 1e-7, tag + " denref")


@pytest.mark.parametrize("form", ["numpy", "c"])
def test_merge_ref_denoiser(golden, form):
    g = golden("merge")
    om = _merge_form(form)
    H, W = g["ref"].shape
    cfg = base_config(ts=16, scale=2)
    cfg.accumulated_robustness_denoiser.enabled = True
    cfg.accumulated_robustness_denoiser.merge.enabled = True
    num, den = acc_pattern(2 * H, 2 * W, 0), acc_pattern(2 * H, 2 * W, 5)
    om.merge_ref(g["ref"], g["covs_ref"], num, den, g["cfa"], cfg, g["acc_rob"].astype(np.float64))
    assert_close(num, g["den_numref"], 1e-5, 1e-7, "denoiser numref")
    assert_close(den, g["den_denref"], 1e-5, 1e-7, "denoiser denref")


def x1_config(cfa, wb):
    cfg = base_config(ts=16, scale=1)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.exif = {"cfa_pattern": [list(r) for r in cfa], "iso": 100, "white_balance": list(wb)}
    cfg.accumulated_robustness_denoiser.enabled = True
    cfg.accumulated_robustness_denoiser.merge.enabled = True
    return cfg


def test_e2e_128(golden):
    """main() on the 128x128 x3 burst the reference itself processed (x2, Ts=16, all-L2)."""
    from handheld_super_resolution import synthetic as synth

    g = golden("e2e_128")
    ref, comp, shifts = synth.make_burst(128, 128, 3, seed=int(g["seed"]), max_shift=2.0, occluder=True)
    np.testing.assert_array_equal(shifts, g["shifts"])
    cfg = base_config(ts=16, scale=2)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.robustness.save_mask = True
    cap = {}
    out, dbg = oracle.main(ref, comp, cfg, capture=cap)
    assert_close(cap["grey_ref"], g["grey_ref"], 0, 2e-6, "grey ref")
    flow = np.stack(cap["flow"])
    # measured: flow 4e-6 px, r 3e-6, output 5e-6 (the sim accumulates in a different float64/float32 mix than
    # NumPy in a few places) — asserted with a 10x margin, no outliers allowed
    assert_close(flow, g["flow"], 0, 5e-5, "flow")
    assert_close(np.stack(cap["r"]), g["r"], 0, 5e-5, "r")
    assert_close(cap["covs"][-1], g["covs_last"], 1e-4, 1e-6, "ref covs")
    assert_close(dbg["accumulated robustness"], g["acc_r"], 0, 5e-5, "acc r")
    assert_close(out, g["out"], 0, 5e-5, "output")


def test_e2e_x1_denoiser(golden):
    """main() in BASELINE config C1's regime as the reference itself computed it: x1, BGGR CFA, white balance
    (1.9, 1, 1.6), accumulated-robustness merge denoiser on; 128x160, 3 frames."""
    from handheld_super_resolution import synthetic as synth

    g = golden("e2e_x1")
    cfa, wb = ((2, 1), (1, 0)), (1.9, 1.0, 1.6)
    ref, comp, shifts = synth.make_burst(128, 160, 3, seed=int(g["seed"]), max_shift=2.0, occluder=True, cfa=cfa, wb=wb)
    np.testing.assert_array_equal(shifts, g["shifts"])
    cfg = x1_config(cfa, wb)
    cap = {}
    out, dbg = oracle.main(ref, comp, cfg, capture=cap)
    assert_close(np.stack(cap["flow"]), g["flow"], 0, 5e-5, "flow")
    assert_close(np.stack(cap["r"]), g["r"], 0, 5e-5, "r")
    assert_close(dbg["accumulated robustness"], g["acc_r"], 0, 5e-5, "acc r")
    assert_close(out, g["out"], 0, 5e-5, "output")


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("tag", ["s15", "s3", "s2iso", "ts32"])
def test_e2e_scales(golden, tag, fast):
    """main() as the reference itself computed it at x1.5 (GRBG, white balance), x3 (4 frames) and x2 with isotropic
    kernels (GBRG, white balance): tools/refsim stage e2e_scales."""
    from helpers import e2e_scales_case

    g = golden("e2e_scales")
    ref, comp, shifts, cfg_fn = e2e_scales_case(tag)
    np.testing.assert_array_equal(shifts, g[f"{tag}_shifts"])
    cap = {}
    out, dbg = oracle.main(ref, comp, cfg_fn(), capture=cap, fast=fast)
    assert_close(np.stack(cap["flow"]), g[f"{tag}_flow"], 0, 5e-5, "flow")
    assert_close(np.stack(cap["r"]), g[f"{tag}_r"], 0, 5e-5, "r")
    assert_close(dbg["accumulated robustness"], g[f"{tag}_acc_r"], 0, 5e-5, "acc r")
    assert_close(out, g[f"{tag}_out"], 0, 5e-5, "output")


def test_post_path_golden(golden):
    """The step after the path: orientation, median frame-count denoiser and raw2rgb.postprocess
    without sharpening against outputs of the reference's own functions (tools/refsim stage "post")."""
    from oracle import post

    g = golden("post")
    for ori in range(1, 9):
        assert np.array_equal(post.apply_orientation(g["ori_in"], ori), g[f"ori{ori}"]), ori
    med = post.frame_count_denoising_median(g["med_in"], g["med_racc"], {"radius_max": 3, "max_frame_count": 8},
                                            int(g["med_scale"]))
    assert np.array_equal(med, g["med_out"])
    assert int(g["gauss_runs_upstream"]) == 0  # range() of a float: the gauss denoiser does not type under Numba
    assert np.allclose(post.get_color_matrix(g["pp_xyz2cam"]), g["pp_ccm"], rtol=0, atol=1e-7)
    off = {"enabled": False}
    kw = dict(do_tonemapping=False, sharpening=off, xyz2cam=g["pp_xyz2cam"])
    assert np.allclose(post.postprocess(g["pp_in"], do_color_correction=False, do_gamma=True, **kw), g["pp_gamma_only"],
                       rtol=0, atol=1e-7)
    assert np.allclose(post.postprocess(g["pp_in"], do_color_correction=True, do_gamma=True, **kw), g["pp_ccm_gamma"],
                       rtol=0, atol=5e-7)
    assert np.allclose(post.postprocess(g["pp_in"], do_color_correction=True, do_gamma=False, do_devignette=True, **kw),
                       g["pp_ccm_devig"], rtol=0, atol=5e-7)
    assert np.allclose(post.postprocess(g["pp_in"], do_color_correction=True, do_tonemapping=False, do_gamma=True,
                                        sharpening=None, xyz2cam=np.zeros((3, 3))), g["pp_zero_ccm"], rtol=0, atol=5e-7)


def test_post_gauss_and_unsharp_known_answers():
    """Oracle-only parts of the post path: the gauss denoiser (upstream cannot run it) and the unsharp mask (restated
    through scipy.ndimage.gaussian_filter, the routine skimage.filters.unsharp_mask calls)."""
    from oracle import post

    rng = np.random.default_rng(0)
    img = rng.random((12, 15, 3)).astype(np.float32)
    # accumulated robustness >= max_frame_count everywhere: sigma = 0 / radius = 0 -> identity
    full = np.full((6, 8), 9.0)
    assert np.array_equal(post.frame_count_denoising_gauss(img, full, {"sigma_max": 1.5, "max_frame_count": 8}, 2), img)
    assert np.array_equal(post.frame_count_denoising_median(img, full, {"radius_max": 3, "max_frame_count": 8}, 2), img)
    # constant image: any normalised blur is the identity (also at the borders)
    const = np.full((9, 11, 3), 0.3, np.float32)
    zero = np.zeros((5, 6))
    assert np.allclose(post.frame_count_denoising_gauss(const, zero, {"sigma_max": 1.5, "max_frame_count": 8}, 2), 0.3, atol=1e-7)
    assert np.allclose(post.unsharp_mask(const, 3, 1.5), 0.3, atol=1e-6)
    # an impulse is sharpened: centre grows by amount * (1 - g0^2), g0 = central tap of the normalised Gaussian
    imp = np.zeros((41, 41, 3), np.float32)
    imp[20, 20] = 1.0
    from handheld_super_resolution.raw2rgb import gaussian_taps

    taps, radius = gaussian_taps(3)
    assert radius == 12 and abs(taps.sum() - 1) < 1e-15
    got = post.unsharp_mask(imp, 3, 1.5)[20, 20, 0]
    assert abs(got - (1 + 1.5 * (1 - taps[radius] ** 2))) < 1e-6


MONO = ((1, 1), (1, 1))  # synthetic "CFA" that samples the scene's green plane at every pixel: a monochrome sensor


def test_grey_mode_stages(golden):
    """`mode: grey` (monochrome sensors): per-pixel covariances, one-channel robustness (including the
    reference's hard-coded s = 2 in the statistics upscale, which stretches the top-left quadrant over the frame) and
    the one-channel merge / merge_ref, against outputs of the reference's own functions (tools/refsim "grey_mode")."""
    g = golden("grey_mode")
    cfa, wb = np.array([[0, 1], [1, 2]]), np.ones(3)
    cfg = base_config(mode="grey")
    covs = oracle.estimate_kernels(g["k_raw"], cfg)
    assert covs.shape == g["k_raw"].shape + (2, 2)
    assert_close(covs, g["k_cov"], 2e-5, 1e-7, "grey covs")
    rm, rv = oracle.init_robustness(g["r_ref"], cfa, wb, cfg)
    assert rm.shape == (1,) + g["r_ref"].shape
    assert_close(rm, g["r_means"], 1e-6, 1e-8, "grey ref means")
    assert_close(rv, g["r_vars"], 1e-5, 1e-9, "grey ref vars")
    std, diff = np.array(cfg.noise_model.std_curve), np.array(cfg.noise_model.diff_curve)
    r = oracle.compute_robustness(g["r_comp"], g["r_means"], g["r_vars"], g["r_flow"], cfa, wb, (std, diff), cfg)
    assert_close(r, g["r_out"], 1e-4, 1e-5, "grey r")
    H, W = g["m_comp"].shape
    for tag, scale, kern in (("s2", 2, "steerable"), ("s15", 1.5, "steerable"), ("s3", 3, "steerable"), ("s2iso", 2, "iso")):
        cfg = base_config(ts=16, scale=scale, mode="grey")
        cfg.merging.kernel = kern
        oh, ow = round(scale * H), round(scale * W)
        num, den = acc_pattern(oh, ow, 0), acc_pattern(oh, ow, 5)
        oracle.merge(g["m_comp"], g["m_flow"], g["m_covs"], g["m_r"], num, den, cfa, cfg)
        assert np.array_equal(num[..., 1:], acc_pattern(oh, ow, 0)[..., 1:])  # channels 1, 2 untouched
        assert_close(num, g[f"m_{tag}_num"], 2e-6, 1e-7, tag + " num")
        assert_close(den, g[f"m_{tag}_den"], 2e-6, 1e-7, tag + " den")
        num, den = acc_pattern(oh, ow, 0), acc_pattern(oh, ow, 5)
        oracle.merge_ref(g["m_ref"], g["m_covs_ref"], num, den, cfa, cfg)
        assert_close(num, g[f"m_{tag}_numref"], 1e-5, 1e-7, tag + " numref")
        assert_close(den, g[f"m_{tag}_denref"], 1e-5, 1e-7, tag + " denref")
    cfg = base_config(ts=16, scale=2, mode="grey")
    cfg.accumulated_robustness_denoiser.enabled = True
    cfg.accumulated_robustness_denoiser.merge.enabled = True
    num, den = acc_pattern(2 * H, 2 * W, 0), acc_pattern(2 * H, 2 * W, 5)
    oracle.merge_ref(g["m_ref"], g["m_covs_ref"], num, den, cfa, cfg, g["m_acc_rob"].astype(np.float64))
    assert_close(num, g["m_den_numref"], 1e-5, 1e-7, "grey denoiser numref")
    assert_close(den, g["m_den_denref"], 1e-5, 1e-7, "grey denoiser denref")


def grey_e2e_inputs(g):
    from handheld_super_resolution import synthetic as synth

    ref, comp, shifts = synth.make_burst(128, 128, 3, seed=int(g["e_seed"]), cfa=MONO, max_shift=2.0, occluder=True)
    np.testing.assert_array_equal(shifts, g["e_shifts"])
    cfg = base_config(ts=16, scale=2, mode="grey")
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.robustness.save_mask = True
    return ref, comp, cfg


def test_e2e_grey_mode(golden):
    """main() with `mode: grey` on the 128x128 x3 monochrome burst the reference itself processed (x2, Ts=16, all-L2):
    channel 0 carries the image, channels 1 and 2 are 0/0 = NaN (the accumulators always have three channels,
    super_resolution.py:123-124)."""
    g = golden("grey_mode")
    ref, comp, cfg = grey_e2e_inputs(g)
    cap = {}
    out, dbg = oracle.main(ref, comp, cfg, capture=cap)
    assert np.isnan(g["e_out"][..., 1:]).all() and np.isfinite(g["e_out"][..., 0]).mean() > 0.99
    assert_close(np.stack(cap["flow"]), g["e_flow"], 0, 5e-5, "flow")
    assert_close(np.stack(cap["r"]), g["e_r"], 0, 5e-5, "r")
    assert_close(dbg["accumulated robustness"], g["e_acc_r"], 0, 5e-5, "acc r")
    assert_close(out, g["e_out"], 0, 5e-5, "output")


SENSORS = ("rggb10", "bggr14", "grbg12")


def test_frontend_golden(golden):
    """oracle.frontend.normalize_burst against what the reference's own load_dng_burst
    (utils_dng.py:50-164) returned for three synthetic sensors (tools/refsim: the decoder is a stand-in, the loader's
    arithmetic, CFA relabelling and ISO clipping are upstream code) — bit for bit."""
    g = golden("frontend")
    for t in SENSORS:
        cfa = g[f"{t}_pattern"].copy()
        cfa[cfa == 3] = 1
        assert np.array_equal(cfa, g[f"{t}_cfa"])
        got = oracle.frontend.normalize_burst(g[f"{t}_counts"], g[f"{t}_black"].tolist(), int(g[f"{t}_white"]),
                                              g[f"{t}_wb"].tolist(), cfa)
        assert got.dtype == np.float32
        assert np.array_equal(got[0], g[f"{t}_ref"]) and np.array_equal(got[1:], g[f"{t}_comp"]), t
        assert int(g[f"{t}_iso"]) == min(3200, max(100, int(g[f"{t}_iso_in"])))
        assert np.array_equal(g[f"{t}_xyz2cam"], g[f"{t}_ccm_in"].reshape(3, 3).astype(np.float32))

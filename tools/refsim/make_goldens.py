"""Capture golden vectors by executing the reference's own code on CPU.

TEST INFRASTRUCTURE ONLY.  Run in the build container (where /root/reference
exists):

    python -m tools.refsim.make_goldens [stage ...]      # default: all stages

It imports the reference through ``tools.refsim.loader`` (fake numba, torch on
CPU), feeds it small seeded inputs and writes ``tests/golden/<stage>.npz``
holding inputs and the reference's outputs.  The reference's sources never
leave /root/reference; only these data files are committed.

The default level-0 metric (L1) is undefined behaviour upstream (SURVEY.md
App. A D1) and cannot be captured; every golden uses L2.
"""
from __future__ import annotations

import copy
import os
import sys
import time

import numpy as np
import torch

from . import loader

GOLDEN = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden"))
# HHSR_REFSIM_AUDIT=1 (tools/refsim/audit.py): the fixtures go to a scratch directory and are compared with the committed
# ones — an audited run must reproduce them bit for bit —, the recorded type classes to tests/golden/typing_audit.json
OUT = os.environ.get("HHSR_GOLDEN_OUT") or (os.path.join("/tmp", "hhsr_golden_audit") if loader.AUDIT_CALLS else GOLDEN)
OUT = os.path.abspath(OUT)

cfgmod = loader.install()
from numba import cuda  # noqa: E402  (the fake one)


def _load_synth():
    import importlib.util

    p = os.path.join(os.path.dirname(GOLDEN), "..", "handheld-multi-frame-super-resolution_amd",
                     "handheld_super_resolution", "synthetic.py")
    spec = importlib.util.spec_from_file_location("_refsim_synth", os.path.abspath(p))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


synth = _load_synth()
ALPHA, BETA = synth.ALPHA_ISO100, synth.BETA_ISO100


def smooth_field(rng, h, w, sigma=2.0, amp=1.0):
    """Band-limited random image in [0,1] (textured enough for LK / matching)."""
    from scipy.ndimage import gaussian_filter

    f = gaussian_filter(rng.standard_normal((h + 16, w + 16)), sigma)
    f = (f - f.min()) / (f.max() - f.min())
    return (amp * f).astype(np.float32)


def base_config(ts=16, scale=2, **kw):
    cfg = cfgmod.default_config()
    cfg.scale = scale
    cfg.verbose = 0
    cfg.block_matching.tuning.tile_size = ts
    cfg.block_matching.tuning.metrics = ["L2", "L2", "L2", "L2"]
    cfg.noise_model.alpha = ALPHA
    cfg.noise_model.beta = BETA
    params = loader.ref("params")
    params.update_snr_config(cfg, 30.0)
    std, diff = synth.noise_curves(ALPHA, BETA)
    cfg.noise_model.update({"std_curve": std.tolist(), "diff_curve": diff.tolist()})
    cfg.exif = {"cfa_pattern": [[0, 1], [1, 2]], "iso": 100, "white_balance": [1.0, 1.0, 1.0]}
    cfg.accumulated_robustness_denoiser.enabled = False
    for k, v in kw.items():
        cfg[k] = v
    return cfg


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    p = os.path.join(OUT, name + ".npz")
    np.savez_compressed(p, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {p} ({os.path.getsize(p) / 1024:.1f} KiB)")


def npy(a):
    if isinstance(a, torch.Tensor):
        return a.detach().contiguous().numpy().copy()
    return np.array(a)


# --------------------------------------------------------------------------- stages
def stage_grey():
    ui = loader.ref("utils_image")
    rng = np.random.default_rng(10)
    out = {}
    for tag, (h, w) in {"a": (48, 40), "b": (50, 42), "c": (31, 45)}.items():
        img = rng.random((h, w), dtype=np.float32)
        out["in_" + tag] = img
        out["out_" + tag] = npy(ui.compute_grey_images(cuda.to_device(img), "FFT"))
    img = rng.random((20, 26), dtype=np.float32)
    out["dec_in"] = img
    out["dec_out"] = npy(ui.compute_grey_images(cuda.to_device(img), "decimating"))
    save("grey", **out)


def stage_downsample():
    ui = loader.ref("utils_image")
    rng = np.random.default_rng(11)
    img = rng.random((70, 83), dtype=np.float32)
    t = torch.as_tensor(img)[None, None]
    o2 = ui.cuda_downsample(t, "gaussian", 2)
    o4 = ui.cuda_downsample(t, "gaussian", 4)
    # a 4-level pyramid (coarse first in the reference's list)
    al = loader.ref("alignment")
    img2 = rng.random((200, 220), dtype=np.float32)
    pyr = al.build_gaussian_pyramid(torch.as_tensor(img2)[None, None], [1, 2, 4, 2])
    save("downsample", img=img, f2=npy(o2.squeeze()), f4=npy(o4.squeeze()), img2=img2,
         pyr0=npy(pyr[3]), pyr1=npy(pyr[2]), pyr2=npy(pyr[1]), pyr3=npy(pyr[0]))


def stage_hessian():
    ica = loader.ref("ICA")
    rng = np.random.default_rng(12)
    lvl = smooth_field(rng, 64, 80)[:64, :80]
    out = {"lvl": lvl}
    for ts in (8, 16, 32, 64):
        gx, gy, H = ica.init_ica(torch.as_tensor(lvl), ts, None)
        out[f"gx"] = npy(gx)
        out[f"gy"] = npy(gy)
        out[f"H{ts}"] = npy(H)
    save("hessian", **out)


def _bm_case(rng, h, w, ts, r, shift, cfg_l, noise=0.01):
    """ref level [h,w] (multiple of ts), moving level slightly smaller (D16)."""
    big = smooth_field(rng, h + 32, w + 32, sigma=1.5)
    ref = big[16 : 16 + h, 16 : 16 + w].copy()
    sy, sx = shift
    mov = big[16 + sy : 16 + sy + h - 3, 16 + sx : 16 + sx + w - 5].copy()
    mov += noise * rng.standard_normal(mov.shape).astype(np.float32)
    return ref, mov


def stage_bm_l2():
    al = loader.ref("alignment")
    bm = loader.ref("block_matching")
    rng = np.random.default_rng(13)
    out = {}
    for tag, ts, r, shift in (("t16", 16, 4, (2, -3)), ("t8", 8, 4, (-1, 2)), ("t32", 32, 2, (1, 1))):
        h, w = 4 * ts, 5 * ts
        ref, mov = _bm_case(rng, h, w, ts, r, shift, None)
        cfg = base_config(ts=16)
        cfg.block_matching.tuning.tile_sizes = [ts, ts, ts, ts]
        cfg.block_matching.tuning.search_radii = [r, r, r, r]
        tiled = torch.as_tensor(ref).unfold(0, ts, ts).unfold(1, ts, ts)
        tiled = torch.nn.functional.pad(tiled, (r, r, r, r), mode="constant", value=0)
        fft = torch.fft.rfft2(tiled, dim=(-2, -1))
        ny, nx = tiled.shape[:2]
        flow0 = (rng.uniform(-1.6, 1.6, (ny, nx, 2))).astype(np.float32)
        flow0[0, 0] = (0.5, -0.5)  # round-half-even probes
        flow0[0, 1] = (1.5, 2.5)
        flow0[1, 0] = (-1.5, -2.5)
        flow0[-1, -1] = (6.0, 5.0)  # pushes the window over the border (clamp-to-edge)
        flow0[0, -1] = (-7.0, -6.0)
        flow = torch.as_tensor(flow0.copy())
        bm.align_lvl_block_matching_L2(tiled, fft, torch.as_tensor(mov), flow, 0, cfg)
        out.update({f"{tag}_ref": ref, f"{tag}_mov": mov, f"{tag}_flow_in": flow0, f"{tag}_flow_out": npy(flow),
                    f"{tag}_ts_r": np.array([ts, r])})
    save("bm_l2", **out)


def stage_ica():
    ica = loader.ref("ICA")
    rng = np.random.default_rng(14)
    out = {}
    for ts in (8, 16, 32, 64):
        ny, nx = (3, 4) if ts <= 16 else (2, 2)
        h, w = ny * ts, nx * ts
        from scipy.ndimage import shift as nd_shift

        big = smooth_field(rng, h + 32, w + 32, sigma=2.5)
        ref = big[16 : 16 + h, 16 : 16 + w].copy()
        movbig = nd_shift(big.astype(np.float64), (0.3, -0.45), order=3, mode="nearest").astype(np.float32)
        mov = movbig[16 : 16 + h - 2, 16 : 16 + w - 3].copy()  # moving level a bit smaller (D16)
        cfg = base_config(ts=16)
        cfg.block_matching.tuning.tile_sizes = [ts, ts, ts, ts]
        gx, gy, H = ica.init_ica(torch.as_tensor(ref), ts, cfg)
        flow0 = rng.uniform(-0.6, 0.6, (ny, nx, 2)).astype(np.float32)
        flow0[0, 0] = (-1.3, -0.7)   # negative: truncation + signed fraction (D11)
        flow0[-1, -1] = (2.4, 1.8)   # samples beyond the moving image
        flow0[0, -1] = (0.0, 0.0)
        flow = torch.as_tensor(flow0.copy())
        t0 = time.time()
        ica.align_lvl_ica(torch.as_tensor(ref), gx, gy, H, torch.as_tensor(mov), flow, 0, cfg)
        print(f"    ica ts={ts}: {time.time() - t0:.1f}s")
        out.update({f"t{ts}_ref": ref, f"t{ts}_mov": mov, f"t{ts}_flow_in": flow0, f"t{ts}_flow_out": npy(flow),
                    f"t{ts}_H": npy(H)})
    save("ica", **out)


def stage_upscale():
    al = loader.ref("alignment")
    rng = np.random.default_rng(15)
    out = {}
    flow = rng.uniform(-3, 3, (5, 7, 2)).astype(np.float32)
    out["flow"] = flow
    for mode in ("nearest", "bilinear", "bicubic"):
        cfg = base_config(ts=16)
        cfg.block_matching.tuning.flow_upscale_mode = mode
        # level 3 -> 2 (ts 8 -> 16, factor 4: repeat 2), level 2 -> 1 (16 -> 16, factor 4: repeat 4)
        o32 = al.upscale_lvl(torch.as_tensor(flow.copy()), (11, 15), 2, cfg)
        o21 = al.upscale_lvl(torch.as_tensor(flow.copy()), (21, 29), 1, cfg)
        out[f"{mode}_l2"] = npy(o32)
        out[f"{mode}_l1"] = npy(o21)
    save("upscale", **out)


def stage_kernels():
    k = loader.ref("kernels")
    rng = np.random.default_rng(16)
    raw = smooth_field(rng, 40, 48, sigma=1.2)[:40, :48]
    raw = np.clip(raw + 0.02 * rng.standard_normal(raw.shape), 0, 1).astype(np.float32)
    raw[:8, :8] = 0.25  # constant block -> zero structure tensor -> NaN covariances (D10)
    out = {"raw": raw}
    for law in ("linear", "hard_threshold"):
        cfg = base_config()
        cfg.merging.selection_law = law
        out["cov_" + law] = npy(k.estimate_kernels(cuda.to_device(raw), cfg))
    cfg = base_config()
    params = loader.ref("params")
    cfg.merging.tuning.update({"k_detail": "SNR_based", "k_denoise": "SNR_based", "D_th": "SNR_based", "D_tr": "SNR_based"})
    params.update_snr_config(cfg, 10.0)
    out["snr10_params"] = np.array([cfg.merging.tuning.k_detail, cfg.merging.tuning.k_denoise,
                                    cfg.merging.tuning.D_th, cfg.merging.tuning.D_tr])
    out["cov_snr10"] = npy(k.estimate_kernels(cuda.to_device(raw), cfg))
    save("kernels", **out)


def _flows(rng, ny, nx, amp=2.0):
    f = rng.uniform(-amp, amp, (ny, nx, 2)).astype(np.float32)
    f[0, 0] = (-3.2, -2.7)
    f[-1, -1] = (4.6, 3.9)
    return f


def stage_robustness():
    rb = loader.ref("robustness")
    rng = np.random.default_rng(17)
    H, W, ts = 64, 80, 16
    wb = [2.0, 1.0, 1.5]
    ref, comp, _ = synth.make_burst(H, W, 2, seed=77, wb=wb, occluder=True, max_shift=1.0)
    comp = comp[0]
    cfa = np.array([[0, 1], [1, 2]])
    out = {"ref": ref, "comp": comp, "wb": np.array(wb), "cfa": cfa}
    cfg = base_config(ts=ts)
    d_cfa, d_wb = cuda.to_device(cfa), cuda.to_device(np.array(wb))
    guide = rb.compute_guide_image(cuda.to_device(ref), d_cfa, d_wb)
    m, v = rb.compute_local_stats(guide)
    out.update(guide=npy(guide), gmeans=npy(m), gvars=npy(v))
    means, stds = rb.init_robustness(cuda.to_device(ref), d_cfa, d_wb, cfg)
    out.update(ref_means=npy(means), ref_vars=npy(stds))
    flow = rng.uniform(-0.4, 0.4, (H // ts, W // ts, 2)).astype(np.float32)
    flow[1, 2] = (1.7, -1.2)   # flow discontinuity -> s1 around it
    flow[-1, -1] = (3.5, 2.5)  # warps beyond the frame -> +inf -> R = 0
    flow[0, 0] = (-2.5, -1.5)
    out["flow"] = flow
    std, diff = synth.noise_curves(ALPHA, BETA)
    out.update(std_curve=std, diff_curve=diff)
    r = rb.compute_robustness(cuda.to_device(comp), means, stds, cuda.to_device(flow), d_cfa, d_wb,
                              (cuda.to_device(std), cuda.to_device(diff)), cfg)
    out["r"] = npy(r)
    # intermediate maps (same calls compute_robustness makes)
    gcomp = rb.compute_guide_image(cuda.to_device(comp), d_cfa, d_wb)
    cm, _ = rb.compute_local_stats(gcomp)
    cmu = rb.upscale_warp_stats(cm, ts, cuda.to_device(flow))
    d_p = rb.compute_dist(means, cmu)
    d_sq, s_sq = rb.apply_noise_model(d_p, means, stds, cuda.to_device(std), cuda.to_device(diff))
    t = cfg.robustness.tuning
    S = rb.compute_s(cuda.to_device(flow), t.Mt, t.s1, t.s2)
    R = rb.robustness_threshold(d_sq, s_sq, S, t.t, ts, True)
    out.update(comp_means_up=npy(cmu), d_sq=npy(d_sq), sigma_sq=npy(s_sq), S=npy(S), R=npy(R))
    save("robustness", **out)


def acc_pattern(oh, ow, phase):
    """Deterministic non-zero initial accumulator content (recomputed by the tests, not stored)."""
    i = np.arange(oh)[:, None, None]
    j = np.arange(ow)[None, :, None]
    c = np.arange(3)[None, None, :]
    return (((i * 7 + j * 13 + c * 3 + phase) % 17) / 17.0 + 0.25).astype(np.float32)


def stage_merge():
    mg = loader.ref("merge")
    k = loader.ref("kernels")
    rng = np.random.default_rng(18)
    H, W, ts = 32, 48, 16
    ref, comp, _ = synth.make_burst(H, W, 2, seed=99, max_shift=1.0)
    comp = comp[0]
    cfa = np.array([[0, 1], [1, 2]])
    out = {"ref": ref, "comp": comp, "cfa": cfa}
    cfg = base_config(ts=ts)
    covs = npy(k.estimate_kernels(cuda.to_device(comp), cfg))
    covs_ref = npy(k.estimate_kernels(cuda.to_device(ref), cfg))
    covs[3, 5] = np.nan  # NaN covariance -> box weights (D10)
    covs_ref[2, 4] = np.nan
    flow = _flows(rng, H // ts, W // ts, amp=1.5)
    r = rng.random((H, W), dtype=np.float32)
    out.update(covs=covs, covs_ref=covs_ref, flow=flow, r=r)
    for tag, scale, kern, do_ref in (("s2", 2, "steerable", True), ("s15", 1.5, "steerable", True),
                                     ("s1", 1, "steerable", True), ("s3", 3, "steerable", False),
                                     ("s2iso", 2, "iso", True)):
        cfg = base_config(ts=ts, scale=scale)
        cfg.merging.kernel = kern
        oh, ow = round(scale * H), round(scale * W)
        num0, den0 = acc_pattern(oh, ow, 0), acc_pattern(oh, ow, 5)
        num, den = cuda.to_device(num0), cuda.to_device(den0)
        t0 = time.time()
        mg.merge(cuda.to_device(comp), cuda.to_device(flow), cuda.to_device(covs), cuda.to_device(r),
                 num, den, cuda.to_device(cfa), cfg)
        out.update({f"{tag}_num": npy(num), f"{tag}_den": npy(den)})
        if do_ref:
            numr, denr = cuda.to_device(num0), cuda.to_device(den0)
            mg.merge_ref(cuda.to_device(ref), cuda.to_device(covs_ref), numr, denr, cuda.to_device(cfa), cfg)
            out.update({f"{tag}_numref": npy(numr), f"{tag}_denref": npy(denr)})
        print(f"    merge {tag}: {time.time() - t0:.1f}s")
    # accumulated-robustness denoiser variant of merge_ref (Alg. 11 widening + overwrite)
    cfg = base_config(ts=ts, scale=2)
    cfg.accumulated_robustness_denoiser.enabled = True
    cfg.accumulated_robustness_denoiser.merge.enabled = True
    acc_rob = (rng.random((H, W)) * 4).astype(np.float32).astype(np.float64)
    oh, ow = 2 * H, 2 * W
    num0, den0 = acc_pattern(oh, ow, 0), acc_pattern(oh, ow, 5)
    numr, denr = cuda.to_device(num0), cuda.to_device(den0)
    mg.merge_ref(cuda.to_device(ref), cuda.to_device(covs_ref), numr, denr, cuda.to_device(cfa), cfg,
                 cuda.to_device(acc_rob))
    out.update(acc_rob=acc_rob.astype(np.float32), den_numref=npy(numr), den_denref=npy(denr))
    save("merge", **out)


def stage_params():
    params = loader.ref("params")
    rows = []
    for snr in (3.0, 6.0, 10.0, 14.0, 14.5, 22.0, 22.5, 27.3, 30.0, 45.0):
        cfg = cfgmod.default_config()
        params.update_snr_config(cfg, snr)
        t = cfg.merging.tuning
        rows.append([snr, cfg.block_matching.tuning.tile_size, *cfg.block_matching.tuning.tile_sizes,
                     t.k_detail, t.k_denoise, t.D_th, t.D_tr])
    save("params", table=np.array(rows, dtype=np.float64))


def stage_e2e():
    """main() end to end: 128x128, 3 frames, x2, Ts=16, factors [1,2,2,2], all-L2."""
    sr = loader.ref("super_resolution")
    H = W = 128
    ref, comp, shifts = synth.make_burst(H, W, 3, seed=1234, max_shift=2.0, occluder=True)
    cfg = base_config(ts=16, scale=2)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.robustness.save_mask = True
    cap = {"flow": [], "r": [], "covs": [], "grey": []}

    def wrap(name, key):
        f = getattr(sr, name)

        def g(*a, **k):
            o = f(*a, **k)
            cap[key].append(npy(o))
            return o

        setattr(sr, name, g)

    wrap("align", "flow")
    wrap("compute_robustness", "r")
    wrap("estimate_kernels", "covs")
    wrap("compute_grey_images", "grey")
    t0 = time.time()
    with np.errstate(all="ignore"):
        out, dbg = sr.main(ref, comp, cfg)
    print(f"    main(): {time.time() - t0:.1f}s")
    save("e2e_128", shifts=shifts, seed=np.array(1234), grey_ref=cap["grey"][0],
         flow=np.stack(cap["flow"]), r=np.stack(cap["r"]), covs_last=cap["covs"][-1],
         out=npy(out), acc_r=np.asarray(dbg["accumulated robustness"], dtype=np.float32))


def stage_e2e_x1():
    """main() end to end in BASELINE config C1's regime: x1 (demosaicking only), BGGR CFA, white balance != 1,
    accumulated-robustness merge denoiser on (merge_ref widens / overwrites where few frames were merged),
    128x160, 3 frames, Ts=16, factors [1,2,2,2], all-L2."""
    sr = loader.ref("super_resolution")
    H, W = 128, 160
    cfa, wb = ((2, 1), (1, 0)), (1.9, 1.0, 1.6)
    ref, comp, shifts = synth.make_burst(H, W, 3, seed=77, max_shift=2.0, occluder=True, cfa=cfa, wb=wb)
    cfg = base_config(ts=16, scale=1)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.exif = {"cfa_pattern": [list(r) for r in cfa], "iso": 100, "white_balance": list(wb)}
    cfg.accumulated_robustness_denoiser.enabled = True
    cfg.accumulated_robustness_denoiser.merge.enabled = True
    cap = {"flow": [], "r": []}

    def wrap(name, key):
        f = getattr(sr, name)

        def g(*a, **k):
            o = f(*a, **k)
            cap[key].append(npy(o))
            return o

        setattr(sr, name, g)

    wrap("align", "flow")
    wrap("compute_robustness", "r")
    t0 = time.time()
    with np.errstate(all="ignore"):
        out, dbg = sr.main(ref, comp, cfg)
    print(f"    main(): {time.time() - t0:.1f}s")
    save("e2e_x1", shifts=shifts, seed=np.array(77), flow=np.stack(cap["flow"]), r=np.stack(cap["r"]),
         out=npy(out), acc_r=np.asarray(dbg["accumulated robustness"], dtype=np.float32))


E2E_SCALES = {  # name: (H, W, frames, seed, scale, cfa, white balance, kernel, tile size)
    "s15": (128, 160, 3, 5, 1.5, ((1, 0), (2, 1)), (2.1, 1.0, 1.4), "steerable", 16),   # GRBG
    "s3": (128, 128, 4, 9, 3, ((0, 1), (1, 2)), (1.0, 1.0, 1.0), "steerable", 16),       # RGGB, 3 compared frames
    "s2iso": (128, 144, 3, 21, 2, ((1, 2), (0, 1)), (1.7, 1.0, 2.2), "iso", 16),         # GBRG, isotropic kernels
    "ts32": (192, 256, 3, 33, 2, ((2, 1), (1, 0)), (1.0, 1.0, 1.0), "steerable", 32),   # BGGR, 32-pixel tiles
}


def e2e_scales_config(scale, cfa, wb, kernel, ts):
    cfg = base_config(ts=ts, scale=scale)
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.exif = {"cfa_pattern": [list(r) for r in cfa], "iso": 100, "white_balance": list(wb)}
    cfg.merging.kernel = kernel
    return cfg


def stage_e2e_scales():
    """main() end to end at the scales / CFA layouts the two other end-to-end goldens do not cover: x1.5 on a GRBG sensor
    with white balance, x3 with three compared frames, x2 with isotropic kernels on a GBRG sensor (Ts=16, factors
    [1,2,2,2], all-L2)."""
    sr = loader.ref("super_resolution")
    out = {}
    for tag, (H, W, n, seed, scale, cfa, wb, kernel, ts) in E2E_SCALES.items():
        ref, comp, shifts = synth.make_burst(H, W, n, seed=seed, max_shift=2.0, occluder=True, cfa=cfa, wb=wb)
        cfg = e2e_scales_config(scale, cfa, wb, kernel, ts)
        cap = {"flow": [], "r": []}
        orig = {}

        def wrap(name, key):
            f = orig[name] = getattr(sr, name)

            def g(*a, **k):
                o = f(*a, **k)
                cap[key].append(npy(o))
                return o

            setattr(sr, name, g)

        wrap("align", "flow")
        wrap("compute_robustness", "r")
        t0 = time.time()
        try:
            with np.errstate(all="ignore"):
                res, dbg = sr.main(ref, comp, cfg)
        finally:
            for name, f in orig.items():
                setattr(sr, name, f)
        print(f"    {tag}: main() {time.time() - t0:.1f}s")
        out.update({f"{tag}_shifts": shifts, f"{tag}_flow": np.stack(cap["flow"]), f"{tag}_r": np.stack(cap["r"]),
                    f"{tag}_out": npy(res),
                    f"{tag}_acc_r": np.asarray(dbg["accumulated robustness"], dtype=np.float32)})
    save("e2e_scales", **out)


def stage_post():
    """The step after the hot path: apply_orientation, the median frame-count denoiser, and
    raw2rgb.postprocess without sharpening (colour matrix, devignetting, gamma) — all executed upstream code.
    The gauss denoiser is recorded as NOT runnable (range() of a float, utils_image.py:214-216)."""
    from handheld_super_resolution import utils_image as ui
    from handheld_super_resolution import raw2rgb

    rng = np.random.default_rng(77)
    out = {}
    img = rng.random((6, 9, 3)).astype(np.float32)
    for ori in range(1, 9):
        out[f"ori{ori}"] = np.ascontiguousarray(ui.apply_orientation(img, ori))
    out["ori_in"] = img
    # median denoiser: 20 x 28 x 3 image, accumulated robustness spanning radius 0 .. 3
    H, W, scale = 20, 28, 2
    noisy = (smooth_field(rng, H, W)[:H, :W, None] + 0.1 * rng.standard_normal((H, W, 3))).astype(np.float32)
    r_acc = np.linspace(0.0, 10.0, (H // scale) * (W // scale)).reshape(H // scale, W // scale)
    r_acc = r_acc[::-1].copy()
    mcfg = cfgmod.Config({"enabled": True, "radius_max": 3, "max_frame_count": 8, "mode": "bayer", "scale": scale})
    med = ui.frame_count_denoising_median(cuda.to_device(noisy), cuda.to_device(r_acc), mcfg)
    out.update(med_in=noisy, med_racc=r_acc, med_out=npy(med.copy_to_host()), med_scale=scale)
    gcfg = cfgmod.Config({"enabled": True, "sigma_max": 1.5, "max_frame_count": 8, "mode": "bayer", "scale": scale})
    try:
        ui.frame_count_denoising_gauss(cuda.to_device(noisy), cuda.to_device(r_acc), gcfg)
        out["gauss_runs_upstream"] = np.array(1)
    except TypeError as e:  # 'float' object cannot be interpreted as an integer
        out["gauss_runs_upstream"] = np.array(0)
        print("  gauss denoiser does not run upstream:", e)
    # postprocess without sharpening
    pimg = (rng.random((10, 14, 3)) * 1.2 - 0.1).astype(np.float32)
    xyz2cam = np.array([[1.0234, -0.2969, -0.2266], [-0.5625, 1.6328, -0.0469], [-0.0703, 0.2188, 0.6406]], np.float32)
    sharp_off = cfgmod.Config({"enabled": False})
    out["pp_in"], out["pp_xyz2cam"] = pimg, xyz2cam
    out["pp_ccm"] = raw2rgb.get_color_matrix(None, xyz2cam)
    out["pp_gamma_only"] = raw2rgb.postprocess(None, pimg.copy(), False, False, True, sharp_off, False, xyz2cam)
    out["pp_ccm_gamma"] = raw2rgb.postprocess(None, pimg.copy(), True, False, True, sharp_off, False, xyz2cam)
    out["pp_ccm_devig"] = raw2rgb.postprocess(None, pimg.copy(), True, False, False, sharp_off, True, xyz2cam)
    out["pp_zero_ccm"] = raw2rgb.postprocess(None, pimg.copy(), True, False, True, None, False, np.zeros((3, 3), np.float32))
    save("post", **out)


def stage_grey_mode():
    """`mode: grey` (monochrome sensors): the reference's own estimate_kernels / init_robustness / compute_robustness /
    merge / merge_ref on one-channel frames, and main() end to end (128x128, 3 frames, x2, Ts=16, all-L2)."""
    k, rb, mg = loader.ref("kernels"), loader.ref("robustness"), loader.ref("merge")
    sr = loader.ref("super_resolution")
    mono = ((1, 1), (1, 1))  # the synthetic scene's green plane at every pixel
    rng = np.random.default_rng(21)
    out = {}
    # --- kernels
    raw = smooth_field(rng, 40, 48, sigma=1.2)[:40, :48]
    raw = np.clip(raw + 0.02 * rng.standard_normal(raw.shape), 0, 1).astype(np.float32)
    raw[:6, :6] = 0.25
    cfg = base_config(mode="grey")
    out.update(k_raw=raw, k_cov=npy(k.estimate_kernels(cuda.to_device(raw), cfg)))
    # --- robustness
    H, W, ts = 64, 80, 16
    ref, comp, _ = synth.make_burst(H, W, 2, seed=78, cfa=mono, occluder=True, max_shift=1.0)
    comp = comp[0]
    cfa = np.array([[0, 1], [1, 2]])
    d_cfa, d_wb = cuda.to_device(cfa), cuda.to_device(np.array([1.0, 1.0, 1.0]))
    cfg = base_config(ts=ts, mode="grey")
    means, stds = rb.init_robustness(cuda.to_device(ref), d_cfa, d_wb, cfg)
    flow = rng.uniform(-0.4, 0.4, (H // ts, W // ts, 2)).astype(np.float32)
    flow[1, 2] = (1.7, -1.2)
    flow[-1, -1] = (3.5, 2.5)
    flow[0, 0] = (-2.5, -1.5)
    std, diff = synth.noise_curves(ALPHA, BETA)
    r = rb.compute_robustness(cuda.to_device(comp), means, stds, cuda.to_device(flow), d_cfa, d_wb,
                              (cuda.to_device(std), cuda.to_device(diff)), cfg)
    out.update(r_ref=ref, r_comp=comp, r_flow=flow, r_means=npy(means), r_vars=npy(stds), r_out=npy(r))
    # --- merge / merge_ref
    H, W = 32, 48
    ref, comp, _ = synth.make_burst(H, W, 2, seed=98, cfa=mono, max_shift=1.0)
    comp = comp[0]
    cfg = base_config(ts=ts, mode="grey")
    covs = npy(k.estimate_kernels(cuda.to_device(comp), cfg))
    covs_ref = npy(k.estimate_kernels(cuda.to_device(ref), cfg))
    covs[3, 5] = np.nan
    mflow = _flows(rng, H // ts, W // ts, amp=1.5)
    mr = rng.random((H, W), dtype=np.float32)
    out.update(m_ref=ref, m_comp=comp, m_covs=covs, m_covs_ref=covs_ref, m_flow=mflow, m_r=mr)
    for tag, scale, kern in (("s2", 2, "steerable"), ("s15", 1.5, "steerable"), ("s3", 3, "steerable"), ("s2iso", 2, "iso")):
        cfg = base_config(ts=ts, scale=scale, mode="grey")
        cfg.merging.kernel = kern
        oh, ow = round(scale * H), round(scale * W)
        num0, den0 = acc_pattern(oh, ow, 0), acc_pattern(oh, ow, 5)
        num, den = cuda.to_device(num0), cuda.to_device(den0)
        mg.merge(cuda.to_device(comp), cuda.to_device(mflow), cuda.to_device(covs), cuda.to_device(mr),
                 num, den, cuda.to_device(cfa), cfg)
        numr, denr = cuda.to_device(num0), cuda.to_device(den0)
        mg.merge_ref(cuda.to_device(ref), cuda.to_device(covs_ref), numr, denr, cuda.to_device(cfa), cfg)
        out.update({f"m_{tag}_num": npy(num), f"m_{tag}_den": npy(den), f"m_{tag}_numref": npy(numr),
                    f"m_{tag}_denref": npy(denr)})
    cfg = base_config(ts=ts, scale=2, mode="grey")
    cfg.accumulated_robustness_denoiser.enabled = True
    cfg.accumulated_robustness_denoiser.merge.enabled = True
    acc_rob = (rng.random((H, W)) * 4).astype(np.float32).astype(np.float64)
    num0, den0 = acc_pattern(2 * H, 2 * W, 0), acc_pattern(2 * H, 2 * W, 5)
    numr, denr = cuda.to_device(num0), cuda.to_device(den0)
    mg.merge_ref(cuda.to_device(ref), cuda.to_device(covs_ref), numr, denr, cuda.to_device(cfa), cfg,
                 cuda.to_device(acc_rob))
    out.update(m_acc_rob=acc_rob.astype(np.float32), m_den_numref=npy(numr), m_den_denref=npy(denr))
    # --- main()
    H = W = 128
    ref, comp, shifts = synth.make_burst(H, W, 3, seed=4321, cfa=mono, max_shift=2.0, occluder=True)
    cfg = base_config(ts=16, scale=2, mode="grey")
    cfg.block_matching.tuning.factors = [1, 2, 2, 2]
    cfg.robustness.save_mask = True
    cap = {"flow": [], "r": []}
    saved = {}

    def wrap(name, key):
        f = saved[name] = getattr(sr, name)

        def g(*a, **kw):
            o = f(*a, **kw)
            cap[key].append(npy(o))
            return o

        setattr(sr, name, g)

    wrap("align", "flow")
    wrap("compute_robustness", "r")
    t0 = time.time()
    try:
        with np.errstate(all="ignore"):
            res, dbg = sr.main(ref, comp, cfg)
    finally:
        for name, f in saved.items():
            setattr(sr, name, f)
    print(f"    main(): {time.time() - t0:.1f}s")
    out.update(e_shifts=shifts, e_seed=np.array(4321), e_flow=np.stack(cap["flow"]), e_r=np.stack(cap["r"]),
               e_out=npy(res), e_acc_r=np.asarray(dbg["accumulated robustness"], dtype=np.float32))
    save("grey_mode", **out)


def stage_frontend():
    """The step before the hot path: the reference's own `load_dng_burst` (utils_dng.py:50-164) executed
    on a synthetic burst.  No .dng decoder exists offline, so `rawpy.imread` / `exifread.process_file` are replaced by
    stand-ins that hand the loader what a decoder would: integer sensor counts, black levels, white level, camera white
    balance, CFA pattern (rawpy's: second green = 3), the EXIF tags it reads.  Everything after that — the loader's
    frame ordering, its integer -> float32 normalisation and white balance arithmetic (:149-160), the ISO clipping, the
    CFA relabelling, the colour-matrix read — is upstream code.  Three sensors: RGGB 10 bit, BGGR 14 bit with unequal
    black levels, GRBG 12 bit with an ISO below the clip."""
    import fractions
    import tempfile
    import types

    import rawpy as rp_stub
    import exifread as er_stub
    from handheld_super_resolution import utils_dng

    rng = np.random.default_rng(321)
    sensors = [
        dict(tag="rggb10", pattern=[[0, 1], [3, 2]], black=[64, 64, 64, 64], white=1023, wb=[1.91, 1.0, 1.57, 0.0], iso=100,
             shape=(4, 36, 52)),
        dict(tag="bggr14", pattern=[[2, 3], [1, 0]], black=[1020, 1024, 1030, 1024], white=16383, wb=[2.2031, 1.0, 1.4297, 1.0],
             iso=800, shape=(3, 30, 44)),
        dict(tag="grbg12", pattern=[[1, 0], [2, 3]], black=[256, 257, 255, 257], white=4095, wb=[1.5, 1.0, 2.0, 1.0], iso=50,
             shape=(2, 24, 40)),
    ]
    out = {}
    for sns in sensors:
        n, H, W = sns["shape"]
        counts = rng.integers(0, sns["white"] + 1, (n, H, W)).astype(np.uint16)
        counts[0, :2, :4] = [[0, sns["white"], sns["black"][0], sns["black"][1]], [1, sns["white"] - 1, 5, 7]]
        ccm = [fractions.Fraction(int(v), 10000) for v in rng.integers(-9000, 18000, 9)]
        with tempfile.TemporaryDirectory() as d:
            paths = []
            for i in range(n):
                pth = os.path.join(d, f"im_{i:02d}.dng")
                open(pth, "wb").write(b"not a real dng: the decoder is a stand-in")
                paths.append(pth)
            order = {os.path.realpath(p_): i for i, p_ in enumerate(paths)}

            class FakeRaw:
                def __init__(self, path):
                    self.raw_image = counts[order[os.path.realpath(path)]]
                    self.white_level = sns["white"]
                    self.black_level_per_channel = list(sns["black"])
                    self.camera_whitebalance = list(sns["wb"])
                    self.raw_pattern = np.array(sns["pattern"], dtype=np.uint8)

                def __enter__(self):
                    return self

                def __exit__(self, *a):
                    return False

            class Tag:
                def __init__(self, values):
                    self.values = values

                def __str__(self):
                    return str(self.values[0])

            class Ratio:
                def __init__(self, f):
                    self.f = f

                def decimal(self):
                    return float(self.f)

            tags = {"Image PhotometricInterpretation": Tag([32803]), "EXIF ISOSpeedRatings": Tag([sns["iso"]]),
                    "Image Tag 0xC621": Tag([Ratio(f) for f in ccm]), "Image Orientation": Tag([1])}
            rp_stub.imread = lambda path: FakeRaw(path)
            er_stub.process_file = lambda f, **kw: dict(tags)
            # the loader takes the files in glob order; make that order the sorted one on every file system
            import glob as _glob
            real_glob = _glob.glob
            _glob.glob = lambda pat, **kw: sorted(real_glob(pat, **kw))
            try:
                ref_raw, raw_comp, iso, _, cfa, xyz2cam, wb, ref_path = utils_dng.load_dng_burst(d)
            finally:
                _glob.glob = real_glob
        t = sns["tag"]
        assert os.path.basename(ref_path) == "im_00.dng" and ref_raw.dtype == np.float32
        out.update({f"{t}_counts": counts, f"{t}_black": np.array(sns["black"]), f"{t}_white": np.array(sns["white"]),
                    f"{t}_wb": np.array(sns["wb"], np.float64), f"{t}_pattern": np.array(sns["pattern"]),
                    f"{t}_iso_in": np.array(sns["iso"]), f"{t}_ref": ref_raw, f"{t}_comp": raw_comp, f"{t}_iso": np.array(iso),
                    f"{t}_cfa": np.asarray(cfa), f"{t}_xyz2cam": xyz2cam,
                    f"{t}_ccm_in": np.array([float(f) for f in ccm], np.float64)})
    save("frontend", **out)


STAGES = {
    "frontend": stage_frontend,
    "grey_mode": stage_grey_mode,
    "post": stage_post,
    "grey": stage_grey, "downsample": stage_downsample, "hessian": stage_hessian, "bm_l2": stage_bm_l2,
    "ica": stage_ica, "upscale": stage_upscale, "kernels": stage_kernels, "robustness": stage_robustness,
    "merge": stage_merge, "params": stage_params, "e2e": stage_e2e, "e2e_x1": stage_e2e_x1,
    "e2e_scales": stage_e2e_scales,
}


def main(argv):
    names = argv or list(STAGES)
    for n in names:
        print(f"[refsim] {n}")
        t0 = time.time()
        with np.errstate(all="ignore"):
            STAGES[n]()
        print(f"  done in {time.time() - t0:.1f}s")
    if loader.AUDIT_CALLS:
        from . import audit

        same = diff = 0
        for n in names:
            for f in sorted(os.listdir(OUT)):
                if not f.endswith(".npz") or not os.path.exists(os.path.join(GOLDEN, f)):
                    continue
                with np.load(os.path.join(OUT, f)) as a, np.load(os.path.join(GOLDEN, f)) as b:
                    for k in a.files:
                        ok = k in b.files and a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and \
                            np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f")
                        same, diff = same + int(ok), diff + int(not ok)
            break  # (every file once)
        # (HHSR_REFSIM_AUDIT=calls: only the max / min / abs calls, with their call sites — minutes instead of hours)
        rec = audit.dump(os.path.join(GOLDEN, "typing_audit.json" if loader.AUDIT else "typing_audit_calls.json"))
        print(f"[refsim] audit: {len(rec['ops'])} operator classes, {len(rec['calls'])} call classes, {len(rec['vars'])} local "
              f"variables with more than one type ({rec['vars_single_type']} with one); fixtures of this audited run vs the "
              f"committed ones: {same} arrays identical, {diff} different")


if __name__ == "__main__":
    main(sys.argv[1:])

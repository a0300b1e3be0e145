"""Oracle: Alg. 1 driver (restates reference super_resolution.py:41-200).  Test infrastructure."""
import numpy as np

from .grey import compute_grey_images
from .align import init_alignment, align
from .robustness import init_robustness, compute_robustness
from .kernels import estimate_kernels
from .merge import merge as _merge, merge_ref as _merge_ref, divide

F32 = np.float32


def main(ref_img, comp_imgs, config, capture=None, fast=False, flows=None, reuse=None, rob=None, acc_rob=None):
    """Returns (output float32[sH, sW, 3] = num/den, debug_dict) like the reference.

    ``capture``: optional dict that receives per-frame intermediates (grey, flow, r, covs, the reference statistics).
    ``fast``: the accumulation through ``oracle.cfast`` (the same operation sequence compiled from C; bit-identical on
    every case the tests compare).
    ``flows``: per-frame flow fields to use INSTEAD of aligning — the two-sided flow injection of the sweeps (the oracle's
    robustness + kernels + merge on the flows of the implementation under test, tests/test_fuzz_parity.py).
    ``rob``: per-frame robustness maps r [H, W] to use INSTEAD of computing them (with ``flows``: the merge alone, on the
    flows and the robustness of the implementation under test — the third comparison of the sweeps).
    ``acc_rob``: the accumulated robustness [H, W] the reference frame's merge reads (merge.py:223-228) INSTEAD of this run's own
    float64 sum — the implementation under test sums its maps in float32, and the denoiser's decisions
    `acc_rob <= / < max_frame_count` compare that sum with an integer.
    ``reuse``: the ``capture`` of an earlier run on the same burst and config: its flow-independent intermediates
    (kernel covariances, reference statistics) are taken over instead of being recomputed."""
    if fast:
        from . import cfast

        merge, merge_ref = cfast.merge, cfast.merge_ref
    else:
        merge, merge_ref = _merge, _merge_ref
    bayer = config.mode == "bayer"  # grey mode: the frames are their own grey images (super_resolution.py:106-109, 144-147)
    ref = np.asarray(ref_img, dtype=F32)
    comp_imgs = np.asarray(comp_imgs, dtype=F32)
    cfa = np.array(config.exif.cfa_pattern)
    wb = np.array(config.exif.white_balance, dtype=np.float64)
    curves = (np.array(config.noise_model.std_curve, np.float64), np.array(config.noise_model.diff_curve, np.float64))
    accumulate_r = bool(config.accumulated_robustness_denoiser.enabled or config.robustness.save_mask)
    debug = {"robustness": [], "flow": []}

    if flows is None:
        grey_ref = compute_grey_images(ref, config.grey_method) if bayer else ref
        pyr, gxs, gys, hs = init_alignment(grey_ref, config)
    else:
        grey_ref = None
    ref_means, ref_vars = reuse["ref_stats"] if reuse else init_robustness(ref, cfa, wb, config)
    H, W = ref.shape
    acc_r = np.zeros((H, W), np.float64) if accumulate_r else None  # float64 upstream (D15)
    s = config.scale
    osz = (round(s * H), round(s * W))
    num = np.zeros((*osz, 3), F32)
    den = np.zeros((*osz, 3), F32)
    if capture is not None:
        capture.update(grey_ref=grey_ref, flow=[], r=[], covs=[], ref_stats=(ref_means, ref_vars))
    for n in range(comp_imgs.shape[0]):
        img = comp_imgs[n]
        if flows is None:
            grey = compute_grey_images(img, config.grey_method) if bayer else img
            flow = align(pyr, gxs, gys, hs, grey, config)
        else:
            flow = np.asarray(flows[n], dtype=F32)
        r = (np.asarray(rob[n], dtype=F32) if rob is not None else
             compute_robustness(img, ref_means, ref_vars, flow, cfa, wb, curves, config))
        if accumulate_r:
            acc_r += r
        covs = reuse["covs"][n] if reuse else estimate_kernels(img, config)
        merge(img, flow, covs, r, num, den, cfa, config)
        if config.debug:
            debug["flow"].append(flow)
            debug["robustness"].append(r)
        if capture is not None:
            capture["flow"].append(flow)
            capture["r"].append(r)
            capture["covs"].append(covs)
    covs = reuse["covs"][comp_imgs.shape[0]] if reuse else estimate_kernels(ref, config)
    if capture is not None:
        capture["covs"].append(covs)
    acc_in = np.asarray(acc_rob, dtype=np.float64) if (acc_rob is not None and accumulate_r) else acc_r
    merge_ref(ref, covs, num, den, cfa, config, acc_in if accumulate_r else None)
    if capture is not None:
        capture["den"] = den.copy()  # accumulated weights before the normalisation (conditioning of num / den)
    divide(num, den)
    if accumulate_r:
        debug["accumulated robustness"] = acc_r
    return num, debug

"""Oracle: Alg. 1 driver (restates reference super_resolution.py:41-200).  Test infrastructure."""
import numpy as np

from .grey import compute_grey_images
from .align import init_alignment, align
from .robustness import init_robustness, compute_robustness
from .kernels import estimate_kernels
from .merge import merge, merge_ref, divide

F32 = np.float32


def main(ref_img, comp_imgs, config, capture=None):
    """Returns (output float32[sH, sW, 3] = num/den, debug_dict) like the reference.

    ``capture``: optional dict that receives per-frame intermediates (grey, flow, r, covs)."""
    bayer = config.mode == "bayer"  # grey mode: the frames are their own grey images (super_resolution.py:106-109, 144-147)
    ref = np.asarray(ref_img, dtype=F32)
    comp_imgs = np.asarray(comp_imgs, dtype=F32)
    cfa = np.array(config.exif.cfa_pattern)
    wb = np.array(config.exif.white_balance, dtype=np.float64)
    curves = (np.array(config.noise_model.std_curve, np.float64), np.array(config.noise_model.diff_curve, np.float64))
    accumulate_r = bool(config.accumulated_robustness_denoiser.enabled or config.robustness.save_mask)
    debug = {"robustness": [], "flow": []}

    grey_ref = compute_grey_images(ref, config.grey_method) if bayer else ref
    pyr, gxs, gys, hs = init_alignment(grey_ref, config)
    ref_means, ref_vars = init_robustness(ref, cfa, wb, config)
    H, W = ref.shape
    acc_r = np.zeros((H, W), np.float64) if accumulate_r else None  # float64 upstream (D15)
    s = config.scale
    osz = (round(s * H), round(s * W))
    num = np.zeros((*osz, 3), F32)
    den = np.zeros((*osz, 3), F32)
    if capture is not None:
        capture.update(grey_ref=grey_ref, flow=[], r=[], covs=[])
    for n in range(comp_imgs.shape[0]):
        img = comp_imgs[n]
        grey = compute_grey_images(img, config.grey_method) if bayer else img
        flow = align(pyr, gxs, gys, hs, grey, config)
        r = compute_robustness(img, ref_means, ref_vars, flow, cfa, wb, curves, config)
        if accumulate_r:
            acc_r += r
        covs = estimate_kernels(img, config)
        merge(img, flow, covs, r, num, den, cfa, config)
        if config.debug:
            debug["flow"].append(flow)
            debug["robustness"].append(r)
        if capture is not None:
            capture["flow"].append(flow)
            capture["r"].append(r)
            capture["covs"].append(covs)
    covs = estimate_kernels(ref, config)
    if capture is not None:
        capture["covs"].append(covs)
    merge_ref(ref, covs, num, den, cfa, config, acc_r if accumulate_r else None)
    if capture is not None:
        capture["den"] = den.copy()  # accumulated weights before the normalisation (conditioning of num / den)
    divide(num, den)
    if accumulate_r:
        debug["accumulated robustness"] = acc_r
    return num, debug

"""Alg. 5 kernel covariances (reference kernels.py)."""
import torch

from . import _lib

SEL_HARD_THRESHOLD = 0
SEL_LINEAR = 1


def _kernel_params(config):
    law = config.merging.selection_law
    if law == "hard_threshold":
        sel = SEL_HARD_THRESHOLD
    elif law == "linear":
        sel = SEL_LINEAR
    else:
        raise ValueError(f"Unknown selection law: {law}")
    t = config.merging.tuning
    alpha, beta = config.noise_model.alpha, config.noise_model.beta
    assert alpha > 0, f"alpha should be positive, got {alpha} (VST is ill defined and kernels would be wrong)"
    return (float(alpha), float(beta), float(t.k_detail), float(t.k_denoise), float(t.D_th), float(t.D_tr),
            float(t.k_stretch), float(t.k_shrink), sel)


def estimate_kernels(img, config):
    """covs float32[H/2, W/2, 2, 2] sampled at the centre of every Bayer quad (kernels.py:29-137).
    GAT, 2x2 decimation, the two gradient convolutions and the per-quad kernel are ONE HIP kernel.
    `mode: grey` (monochrome sensors): no decimation, covs float32[H, W, 2, 2], one per pixel (kernels.py:83-87)."""
    params = _kernel_params(config)
    img = _lib.f32c(img)
    H, W = img.shape
    if config.mode != "bayer":
        return mono_frame_stats(img, config, stats=False)[2]
    if H % 2 or W % 2:
        raise ValueError(f"bayer frames need even dimensions, got {(H, W)}")
    covs = torch.empty((H // 2, W // 2, 2, 2), dtype=torch.float32, device=img.device)
    _lib.call("hhsr_cov_from_raw", _lib.ptr(img), H, W, W, _lib.ptr(covs), *params, _lib.stream())
    return covs


def frame_stats(img, cfa_pattern, white_balance, config, want_vars=False):
    """One pass over the raw frame for both of its per-frame consumers: the guide-image local means (and,
    for the reference frame, variances) of robustness.py:207-294 and the kernel covariances of kernels.py:29-137.
    Returns (means [3, H/2, W/2], vars or None, covs [H/2, W/2, 2, 2])."""
    from .robustness import _wb3

    if config.mode != "bayer":
        return mono_frame_stats(img, config, want_vars=want_vars)
    params = _kernel_params(config)
    img = _lib.f32c(img)
    H, W = img.shape
    if H % 2 or W % 2:
        raise ValueError(f"bayer frames need even dimensions, got {(H, W)}")
    means = torch.empty((3, H // 2, W // 2), dtype=torch.float32, device=img.device)
    vars_ = torch.empty_like(means) if want_vars else None
    covs = torch.empty((H // 2, W // 2, 2, 2), dtype=torch.float32, device=img.device)
    _lib.call("hhsr_frame_stats", _lib.ptr(img), H, W, W, _lib.cfa_bytes(cfa_pattern),
              _lib.doubles(_wb3(white_balance)), _lib.ptr(means), _lib.ptr(vars_), _lib.ptr(covs), *params,
              _lib.stream())
    return means, vars_, covs


def frame_stats_batch(imgs, cfa_pattern, white_balance, config):
    """frame_stats() of several comp frames of one shape in one launch (hhsr_frame_stats_batch): list of
    (means [3, H/2, W/2], None, covs [H/2, W/2, 2, 2]); per frame bit-identical."""
    from .robustness import _wb3

    params = _kernel_params(config)
    imgs = [_lib.f32c(i) for i in imgs]
    H, W = imgs[0].shape
    if H % 2 or W % 2:
        raise ValueError(f"bayer frames need even dimensions, got {(H, W)}")
    n, dev = len(imgs), imgs[0].device
    means = list(torch.empty((n, 3, H // 2, W // 2), dtype=torch.float32, device=dev).unbind(0))
    covs = list(torch.empty((n, H // 2, W // 2, 2, 2), dtype=torch.float32, device=dev).unbind(0))
    _lib.call("hhsr_frame_stats_batch", _lib.ptr_array(imgs), n, H, W, W, _lib.cfa_bytes(cfa_pattern),
              _lib.doubles(_wb3(white_balance)), _lib.ptr_array(means), _lib.ptr_array(covs), *params, _lib.stream())
    return [(m, None, c) for m, c in zip(means, covs)]


def mono_frame_stats(img, config, stats=True, covs=True, want_vars=False):
    """`mode: grey`: the per-frame pass of a monochrome frame — 3x3 local means (and variances) of the frame itself
    [1, H, W] (robustness.py:62-66, 269-294) and / or the per-pixel kernel covariances [H, W, 2, 2] (kernels.py:83-137)."""
    img = _lib.f32c(img)
    H, W = img.shape
    params = _kernel_params(config) if covs else (1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0)
    means = torch.empty((1, H, W), dtype=torch.float32, device=img.device) if stats else None
    vars_ = torch.empty_like(means) if stats and want_vars else None
    cov = torch.empty((H, W, 2, 2), dtype=torch.float32, device=img.device) if covs else None
    _lib.call("hhsr_mono_frame_stats", _lib.ptr(img), H, W, W, _lib.ptr(means), _lib.ptr(vars_), _lib.ptr(cov), *params,
              _lib.stream())
    return means, vars_, cov

"""SNR-derived parameters and config checks (reference params.py:4-123); host-side, tiny."""
import numpy as np


def lerp(x, x_range, y_range):
    x0, x1 = x_range
    y0, y1 = y_range
    assert x0 < x1
    assert y0 != y1
    t = (x - x0) / (x1 - x0)
    t = max(0.0, min(1.0, t))
    return y0 + (y1 - y0) * t


def update_snr_config(config, SNR):
    """SNR -> tile size (<=14: 64, <=22: 32, else 16), tile_sizes and the merge tunings, in place."""
    SNR = float(np.clip(SNR, 6, 30))
    Ts = 64 if SNR <= 14 else (32 if SNR <= 22 else 16)
    bm = config.block_matching.tuning
    if bm.tile_size != "SNR_based":
        assert isinstance(bm.tile_size, int), "tile_size should be an integer or 'SNR_based'"
        Ts = bm.tile_size
    else:
        bm.tile_size = Ts
    bm.tile_sizes = [int(Ts * s) for s in bm.tile_size_factors]
    t = config.merging.tuning
    for key, yr in (("k_detail", [0.33, 0.25]), ("k_denoise", [5.0, 3.0]), ("D_th", [0.81, 0.71]), ("D_tr", [1.24, 1])):
        if t[key] == "SNR_based":
            t[key] = lerp(SNR, [6, 30], yr)
        else:
            assert isinstance(t[key], float), f"{key} should be a float or 'SNR_based'"


def level_shapes(imshape, config):
    """True pyramid level shapes (reference side is circularly padded to k*Ts, the moving side is not —
    SURVEY.md App. D) and the tile grids, fine-to-coarse."""
    bm = config.block_matching.tuning
    Ts, factors, tss = bm.tile_size, bm.factors, bm.tile_sizes
    h, w = imshape
    ph = h + (Ts - h % Ts) * (h % Ts != 0)
    pw = w + (Ts - w % Ts) * (w % Ts != 0)
    ref, mov = [], []
    rs, ms = (ph, pw), (h, w)
    for f in factors:
        if f != 1:
            r = int(4 * f * 0.5 + 0.5)
            rs = ((rs[0] - 2 * r) // f, (rs[1] - 2 * r) // f)
            ms = ((ms[0] - 2 * r) // f, (ms[1] - 2 * r) // f)
        ref.append(rs)
        mov.append(ms)
    tiles = [(s[0] // ts, s[1] // ts) for s, ts in zip(ref, tss)]
    return ref, mov, tiles


def sanitize_config(config, imshape):
    """The reference's checks (params.py:4-57) — except that the block-matching feasibility test uses
    the TRUE level shapes (valid convolutions shrink the levels more than floor(prev/f); the reference's
    estimate lets small images through that then get 0x0 tile grids)."""
    if config.mode == "grey" and config.grey_method != "FFT":
        raise NotImplementedError("Grey level images should be obtained with FFT")
    assert config.scale >= 1
    den = config.accumulated_robustness_denoiser
    if not config.robustness.enabled and (den.median.enabled or den.gauss.enabled or den.merge.enabled):
        raise ValueError("Accumulated robustness denoiser cannot be enabled if robustness is disabled.")
    if not config.robustness.enabled and config.robustness.save_mask:
        raise ValueError("Robustness mask cannot be saved if robustness is disabled.")
    assert config.merging.kernel in ["steerable", "iso"], f"Unknown kernel type {config.merging.kernel}"
    assert config.mode in ["bayer", "grey"], f"Unknown mode {config.mode}"
    if sum(1 if x.enabled else 0 for x in (den.median, den.gauss, den.merge)) > 1:
        raise ValueError("Only one accumulated robustness denoiser can be enabled at a time.")
    assert config.ica.tuning.n_iter > 0, "Number of ICA iterations should be positive."
    assert config.ica.tuning.sigma_blur >= 0, f"Invalid sigma blur {config.ica.tuning.sigma_blur}."
    assert len(imshape) == 2, f"Input image shape should be 2D, got {imshape}."
    ref, mov, tiles = level_shapes(imshape, config)
    for lvl, (rs, ms, tl) in enumerate(zip(ref, mov, tiles)):
        if min(rs) < 1 or min(ms) < 1 or tl[0] < 1 or tl[1] < 1:
            raise ValueError("Image of shape {} is incompatible with the given block matching tile sizes and "
                             "factors : at level {}, coarse image of shape {} cannot be divided into tiles of "
                             "size {}.".format(imshape, lvl, rs, config.block_matching.tuning.tile_sizes[lvl]))
    valid = ["nearest", "bilinear", "bicubic"]
    assert config.block_matching.tuning.flow_upscale_mode in valid, \
        f"Unknown flow upscaling mode {config.block_matching.tuning.flow_upscale_mode}, should be one of {valid}."

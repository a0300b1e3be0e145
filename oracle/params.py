"""Oracle: SNR-derived parameters (restates reference params.py:4-123).  Test infrastructure."""
import numpy as np


def lerp(x, x_range, y_range):
    """reference params.py:96-123."""
    x0, x1 = x_range
    y0, y1 = y_range
    assert x0 < x1 and y0 != y1
    t = (x - x0) / (x1 - x0)
    t = max(0.0, min(1.0, t))
    return y0 + (y1 - y0) * t


def update_snr_config(config, SNR):
    """reference params.py:59-93: SNR -> tile size and merge tunings (in place)."""
    SNR = float(np.clip(SNR, 6, 30))
    Ts = 64 if SNR <= 14 else (32 if SNR <= 22 else 16)
    bm = config.block_matching.tuning
    if bm.tile_size != "SNR_based":
        assert isinstance(bm.tile_size, int)
        Ts = bm.tile_size
    else:
        bm.tile_size = Ts
    bm.tile_sizes = [int(Ts * s) for s in bm.tile_size_factors]
    t = config.merging.tuning
    for key, yr in (("k_detail", [0.33, 0.25]), ("k_denoise", [5.0, 3.0]), ("D_th", [0.81, 0.71]), ("D_tr", [1.24, 1])):
        if t[key] == "SNR_based":
            t[key] = lerp(SNR, [6, 30], yr)
        else:
            assert isinstance(t[key], float)


def sanitize_config(config, imshape):
    """reference params.py:4-57 (the checks that matter for the hot path)."""
    assert config.scale >= 1
    assert config.merging.kernel in ("steerable", "iso")
    assert config.mode in ("bayer", "grey")
    if not config.robustness.enabled and config.robustness.save_mask:
        raise ValueError("Robustness mask cannot be saved if robustness is disabled.")
    assert config.ica.tuning.n_iter > 0
    assert len(imshape) == 2
    assert config.block_matching.tuning.flow_upscale_mode in ("nearest", "bilinear", "bicubic")

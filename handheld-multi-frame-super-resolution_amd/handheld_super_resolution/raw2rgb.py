"""raw2rgb.postprocess on the device (reference raw2rgb.py:113-128, 198-250): colour matrix, unsharp mask, devignetting,
gamma — and the EXIF orientation folded into the final store — in two passes over the merged image in HBM instead of a
round trip through host NumPy / skimage.  Tone mapping (OpenCV's MergeMertens, raw2rgb.py:163-180) is out of scope."""
import numpy as np
import torch

from . import _lib

RGB2XYZ = np.array([[0.4124564, 0.3575761, 0.1804375],
                    [0.2126729, 0.7151522, 0.0721750],
                    [0.0193339, 0.1191920, 0.9503041]])
_taps_cache = {}


def get_color_matrix(raw, xyz2cam=None):
    """rgb2cam, rows normalised (raw2rgb.py:113-128).  `raw`: a rawpy image (its rgb_xyz_matrix is the fallback) or None."""
    if xyz2cam is None:
        if raw is None:
            raise ValueError("colour correction needs xyz2cam (the DNG ColorMatrix1) or a rawpy image")
        xyz2cam = raw.rgb_xyz_matrix[:3]
    xyz2cam = np.asarray(xyz2cam, np.float64)
    if np.linalg.norm(xyz2cam) == 0:
        print("Warning -- CCM not found or given. Use eye matrix instead.")
        rgb2cam = RGB2XYZ
    else:
        rgb2cam = xyz2cam @ RGB2XYZ
    rgb2cam = rgb2cam / rgb2cam.sum(axis=-1, keepdims=True)
    return rgb2cam.astype(np.float32)


def gaussian_taps(sigma, truncate=4.0):
    """scipy.ndimage's normalised 1-D Gaussian (float64), radius int(truncate sigma + 0.5) — what
    skimage.filters.unsharp_mask blurs with (gaussian(image, sigma=radius, mode='reflect'))."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    phi = np.exp(-0.5 / (float(sigma) * float(sigma)) * x ** 2)
    return phi / phi.sum(), radius


def _device_taps(sigma, device):
    key = (float(sigma), device.index)
    if key not in _taps_cache:
        taps, radius = gaussian_taps(sigma)
        _taps_cache[key] = (torch.as_tensor(taps, dtype=torch.float64, device=device), radius)
    return _taps_cache[key]


def postprocess(raw, img=None, do_color_correction=True, do_tonemapping=True, do_gamma=True, sharpening_config=None,
                do_devignette=False, xyz2cam=None, orientation=1):
    """Same arguments as the reference (raw2rgb.py:206) plus `orientation` (EXIF 1..8, applied in the same pass; the
    reference orients afterwards on the host, super_resolution.py:346-354).  `img`: float32 [H, W, 3] (GPU tensor or
    array); returns a GPU tensor, [H, W, 3] or [W, H, 3] for orientations 5-8."""
    if img is None:
        raise NotImplementedError("rawpy's own pipeline (postprocess(raw) without an image) needs rawpy")
    if do_tonemapping:
        raise NotImplementedError("tone mapping (OpenCV MergeMertens, raw2rgb.py:163-180) is outside the MI355X build")
    img = _lib.f32c(img)
    H, W, C = img.shape
    assert C == 3
    ccm = None
    if do_color_correction:
        ccm = _lib.floats(np.linalg.inv(get_color_matrix(raw, xyz2cam)).astype(np.float32).ravel())
    sharpen = sharpening_config is not None and bool(sharpening_config.get("enabled", False))
    radius, amount = 3, 0.5  # the reference's fall-back when the keys are missing (raw2rgb.py:231-234)
    if sharpen and "radius" in sharpening_config and "amount" in sharpening_config:
        radius, amount = sharpening_config["radius"], sharpening_config["amount"]
    taps, tr, tmp = None, 0, None
    if sharpen:
        taps, tr = _device_taps(radius, img.device)
        tmp = torch.empty_like(img)
    ori = int(orientation)
    out = torch.empty((W, H, 3) if ori >= 5 else (H, W, 3), dtype=torch.float32, device=img.device)
    _lib.call("hhsr_postprocess", _lib.ptr(img), _lib.ptr(tmp), _lib.ptr(out), H, W, ccm, 1 if sharpen else 0,
              float(amount), _lib.ptr(taps), int(tr), 1 if do_devignette else 0, 1 if do_gamma else 0, ori,
              _lib.stream(img.device))
    return out

"""Grey image (Alg. 3) and the decimating Gaussian (reference utils_image.py:58-112, 360-391)."""
import numpy as np
import torch

from . import _lib


class _GreyPlan:
    """Owner of one C-side hipFFT plan (H, W, device); destroyed with the object."""

    def __init__(self, H, W):
        import ctypes
        import os

        h = ctypes.c_void_p()
        _lib.call("hhsr_grey_plan_create", H, W, int(os.environ.get("HHSR_GREY_PLAN", "4")), ctypes.byref(h))
        self.handle = h

    def __del__(self):
        try:
            if self.handle:
                _lib.load().hhsr_grey_plan_destroy(self.handle)
        except Exception:
            pass


_grey_plans = {}  # (H, W, device, stream) -> plan, least recently used first
_GREY_PLAN_CACHE = 16  # e.g. 4 streams x 2 image sizes (bench.py: full size, then the parity crop) x 2 devices


def _grey_plan(H, W, device):
    key = (H, W, device.index, torch.cuda.current_stream(device).cuda_stream)
    p = _grey_plans.pop(key, None)
    if p is None:
        while len(_grey_plans) >= _GREY_PLAN_CACHE:  # evict the least recently used plan only
            _grey_plans.pop(next(iter(_grey_plans)))
        with torch.cuda.device(device):
            p = _GreyPlan(H, W)
    _grey_plans[key] = p  # most recently used last
    return p


def compute_grey_images(img, method):
    """raw -> grey.  "FFT": ideal half-band low-pass (utils_image.py:82-100).  The reference runs a full
    complex FFT, zeroes the outer quarter bands of the shifted spectrum and keeps the real part; here the
    spectrum is the half spectrum of rocFFT's real transform and the zeroing is the equivalent Hermitian
    mask applied in place by a HIP kernel (no fftshift copies), so irfft2 returns the same real image.
    "decimating": 2x2 mean (utils_image.py:101-112) — only used inside the fused covariance kernel."""
    img = _lib.f32c(img)
    H, W = img.shape
    if method == "FFT":
        # planned rocFFT round trip inside libhhsr_hip.so: r2c -> Hermitian mask (+ normalisation) -> c2r
        out = torch.empty_like(img)
        _lib.call("hhsr_grey_lowpass", _grey_plan(H, W, img.device).handle, _lib.ptr(img), _lib.ptr(out),
                  _lib.stream(img.device))
        return out
    if method == "FFT_torch":  # same maths through torch.fft (rocFFT behind torch), kept for cross-checking
        spec = torch.fft.rfft2(img)
        _lib.call("hhsr_lowpass_mask_r2c", _lib.ptr(spec), H, W, spec.stride(0), spec.stride(1), _lib.stream())
        return torch.fft.irfft2(spec, s=(H, W))
    if method == "FFT_c2c":  # the reference's literal formulation, kept for cross-checking the mask
        spec = torch.fft.fft2(img)
        _lib.call("hhsr_lowpass_mask_c2c", _lib.ptr(spec), H, W, spec.stride(0), spec.stride(1), _lib.stream())
        return torch.fft.ifft2(spec).real.contiguous()
    if method == "decimating":
        h, w = H // 2, W // 2
        v = img[: 2 * h, : 2 * w].double()
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2]) / 4).float()
    raise NotImplementedError("Computation of gray level on GPU is only supported for FFT")


def gaussian_taps(factor):
    """scipy.ndimage's 1-D Gaussian for sigma = factor/2, radius = int(2*factor + 0.5), as the reference
    requests it (utils_image.py:380), restated from its definition; float32 like the reference's tensor."""
    sigma = factor * 0.5
    radius = int(4 * factor * 0.5 + 0.5)
    x = np.arange(-radius, radius + 1, dtype=np.float64)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return (phi / phi.sum()).astype(np.float32)


def cuda_downsample(th_img, kernel="gaussian", factor=2):
    """Valid separable Gaussian + decimation (utils_image.py:360-391) as ONE HIP kernel that writes a
    compact level (the reference returns a strided view of the filtered full-resolution image).
    Accepts [H, W] or [1, 1, H, W] like the reference's callers."""
    if factor == 1:
        return th_img
    if kernel != "gaussian":
        raise ValueError("please use gaussian kernel")
    lead = th_img.shape[:-2]
    img = _lib.f32c(th_img.reshape(th_img.shape[-2:]))
    H, W = img.shape
    taps = gaussian_taps(factor)
    r = (len(taps) - 1) // 2
    h2, w2 = (H - 2 * r) // factor, (W - 2 * r) // factor
    if h2 < 1 or w2 < 1:
        raise ValueError(f"image of shape {(H, W)} is too small to be downsampled by {factor}")
    out = torch.empty((h2, w2), dtype=torch.float32, device=img.device)
    _lib.call("hhsr_gauss_decimate", _lib.ptr(img), H, W, W, _lib.ptr(out), w2, factor, _lib.floats(taps),
              len(taps), _lib.stream())
    return out.reshape(*lead, h2, w2)

"""Grey image (Alg. 3) and the decimating Gaussian (reference utils_image.py:58-112, 360-391)."""
import numpy as np
import torch

from . import _lib


class _GreyPlan:
    """Owner of one C-side hipFFT plan (H, W, device); destroyed with the object.  `batch`: spectra the plan holds =
    frames one launch per phase may carry (HHSR_GREY_BATCH)."""

    def __init__(self, H, W, batch=1):
        import ctypes
        import os

        _free_deferred_plans()
        h = ctypes.c_void_p()
        _lib.call("hhsr_grey_plan_create", H, W, int(os.environ.get("HHSR_GREY_PLAN", "4")) | (int(batch) << 8),
                  ctypes.byref(h))
        self.handle = h
        self.batch = int(batch)

    def __del__(self):
        # A plan owns hipMalloc'd buffers: destroying it calls hipFree, which is not permitted while the calling thread
        # captures a stream — and the garbage collector can run THIS finaliser in the middle of somebody's capture (an
        # engine dropped earlier whose graphs kept the plan alive: seen as "operation not permitted when stream is
        # capturing" + abort in tools/debug/emulate_ranks.py).  During a capture the handle is parked and freed later.
        try:
            if not self.handle:
                return
            h, self.handle = self.handle, None
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                _deferred_plans.append(h)
            else:
                _lib.load().hhsr_grey_plan_destroy(h)
        except Exception:
            pass


_deferred_plans = []  # handles of plans whose owner died during a stream capture


def _free_deferred_plans():
    """Destroy the parked plans (called where plans are created: never during a capture)."""
    if _deferred_plans and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        while _deferred_plans:
            try:
                _lib.load().hhsr_grey_plan_destroy(_deferred_plans.pop())
            except Exception:
                pass


_grey_plans = {}  # (H, W, device, stream) -> plan, least recently used first
_GREY_PLAN_CACHE = 16  # e.g. 4 streams x 2 image sizes (bench.py: full size, then the parity crop) x 2 devices


def _grey_plan(H, W, device, batch=1):
    """The plan of (size, device, stream); one that holds fewer spectra than `batch` is replaced by a larger one."""
    key = (H, W, device.index, torch.cuda.current_stream(device).cuda_stream)
    p = _grey_plans.pop(key, None)
    if p is not None and p.batch < batch:
        p = None
    if p is None:
        if torch.cuda.is_current_stream_capturing():
            # a plan allocates (twiddle tables, spectra): it has to exist before its stream is captured — graph.py's
            # runners do one eager call on the capture streams first
            raise RuntimeError(f"no FFT plan for {H}x{W} x{batch} on this stream yet: cannot create one during stream capture")
        while len(_grey_plans) >= _GREY_PLAN_CACHE:  # evict the least recently used plan only
            _grey_plans.pop(next(iter(_grey_plans)))
        with torch.cuda.device(device):
            p = _GreyPlan(H, W, batch)
    _grey_plans[key] = p  # most recently used last
    return p


def compute_grey_images_batch(imgs, method="FFT"):
    """compute_grey_images() of several frames of one shape: ONE launch per transform phase for up to
    _lib.MAX_BATCH frames (hhsr_grey_lowpass_batch) instead of three per frame — the per-frame launches are
    latency-bound (row / column transforms that sit at barriers half of the time), a chunk of frames keeps one
    resident round of workgroups busy across the frames' row blocks.  Per frame bit-identical.  Returns a list of
    [H, W] views of one [n, H, W] tensor."""
    imgs = [_lib.f32c(i) for i in imgs]
    if method != "FFT" or len(imgs) < 2:
        return [compute_grey_images(i, method) for i in imgs]
    H, W = imgs[0].shape
    dev = imgs[0].device
    out = torch.empty((len(imgs), H, W), dtype=torch.float32, device=dev)
    outs = list(out.unbind(0))
    # (a plan that serves more than one frame per launch always holds MAX_BATCH spectra — 24 MB each at 12 MP: the
    # callers' chunk sizes vary, e.g. graph.host_chunks, and a plan cannot be replaced while its stream is captured)
    plan = _grey_plan(H, W, dev, _lib.MAX_BATCH)
    _lib.call("hhsr_grey_lowpass_batch", plan.handle, _lib.ptr_array(imgs), _lib.ptr_array(outs), len(imgs),
              _lib.stream(dev))
    return outs


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


_taps_c = {}  # factor -> (ctypes float array, number of taps): built once, passed by pointer at every launch


def _taps_for_launch(factor):
    hit = _taps_c.get(factor)
    if hit is None:
        taps = gaussian_taps(factor)
        hit = _taps_c[factor] = (_lib.floats(taps), len(taps))
    return hit


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
    taps, ntaps = _taps_for_launch(factor)
    r = (ntaps - 1) // 2
    h2, w2 = (H - 2 * r) // factor, (W - 2 * r) // factor
    if h2 < 1 or w2 < 1:
        raise ValueError(f"image of shape {(H, W)} is too small to be downsampled by {factor}")
    out = torch.empty((h2, w2), dtype=torch.float32, device=img.device)
    _lib.call("hhsr_gauss_decimate", _lib.ptr(img), H, W, W, _lib.ptr(out), w2, factor, taps, ntaps, _lib.stream())
    return out.reshape(*lead, h2, w2)


def cuda_downsample_batch(imgs, factor=2):
    """cuda_downsample() of several [H, W] levels of one shape in one launch (hhsr_gauss_decimate_batch); per frame
    bit-identical.  Returns a list of views of one [n, h2, w2] tensor."""
    if factor == 1:
        return list(imgs)
    imgs = [_lib.f32c(i) for i in imgs]
    H, W = imgs[0].shape
    taps, ntaps = _taps_for_launch(factor)
    r = (ntaps - 1) // 2
    h2, w2 = (H - 2 * r) // factor, (W - 2 * r) // factor
    if h2 < 1 or w2 < 1:
        raise ValueError(f"image of shape {(H, W)} is too small to be downsampled by {factor}")
    out = torch.empty((len(imgs), h2, w2), dtype=torch.float32, device=imgs[0].device)
    outs = list(out.unbind(0))
    _lib.call("hhsr_gauss_decimate_batch", _lib.ptr_array(imgs), len(imgs), H, W, W, _lib.ptr_array(outs), w2, factor,
              taps, ntaps, _lib.stream())
    return outs


# ---- after the path ------------------------------------------------------------------------------------
def apply_orientation(img, ori):
    """EXIF orientation 1..8 (utils_image.py:12-55).  NumPy arrays are handled like upstream; GPU tensors stay on the
    device ([H, W] planes through hhsr_orient_plane, images through flips / transposes of the same memory)."""
    ori = int(ori)
    if not torch.is_tensor(img):
        if ori == 2:
            img = np.flip(img, axis=1)
        elif ori == 3:
            img = np.rot90(img, k=2, axes=(0, 1))
        elif ori == 4:
            img = np.flip(img, axis=0)
        elif ori == 5:
            img = np.rot90(np.flip(img, axis=1), k=-3, axes=(0, 1))
        elif ori == 6:
            img = np.rot90(img, k=-1, axes=(0, 1))
        elif ori == 7:
            img = np.rot90(np.flip(img, axis=1), k=-1, axes=(0, 1))
        elif ori == 8:
            img = np.rot90(img, k=-3, axes=(0, 1))
        return img
    if ori == 1:
        return img
    if img.dim() == 2 and img.is_cuda:
        src = _lib.f32c(img)
        H, W = src.shape
        out = torch.empty((W, H) if ori >= 5 else (H, W), dtype=torch.float32, device=src.device)
        _lib.call("hhsr_orient_plane", _lib.ptr(src), _lib.ptr(out), H, W, ori, _lib.stream(src.device))
        return out
    t = {2: lambda a: a.flip(1), 3: lambda a: a.flip(0).flip(1), 4: lambda a: a.flip(0),
         5: lambda a: a.transpose(0, 1), 6: lambda a: a.transpose(0, 1).flip(1),
         7: lambda a: a.flip(0).flip(1).transpose(0, 1), 8: lambda a: a.transpose(0, 1).flip(0)}[ori]
    return t(img).contiguous()


def _frame_count_denoise(image, r_acc, config, kind, strength, scale, half_index, mode=None):
    # `mode: grey` (monochrome sensors): the reference indexes the accumulated robustness with int(round(y / scale))
    # (utils_image.py:203-204, 260-261) instead of the Bayer branch's half-resolution index.  Upstream reads `mode` (and
    # `scale`) from the denoiser's own sub-block, which process() never fills: here process() passes both.
    mode = str(config.get("mode", "bayer") if mode is None else mode)
    if mode == "grey":
        half_index = 2
    scale = config.get("scale", scale)
    if scale is None:
        raise ValueError("the frame-count denoisers need the scale (config.scale or the scale argument)")
    img = _lib.f32c(image)
    acc = _lib.f32c(r_acc, img.device)
    H, W, C = img.shape
    assert C == 3
    out = torch.empty_like(img)
    _lib.call("hhsr_frame_count_denoise", _lib.ptr(img), _lib.ptr(out), H, W, _lib.ptr(acc), acc.shape[0], acc.shape[1],
              float(scale), kind, float(strength), float(config.max_frame_count), int(half_index),
              _lib.stream(img.device))
    return out


def frame_count_denoising_median(image, r_acc, config, scale=None, half_index=True, mode=None):
    """Median filter whose radius grows where few frames were merged (utils_image.py:236-286).  `config`: the
    accumulated_robustness_denoiser.median block (radius_max <= 7, max_frame_count); the reference reads `scale` and
    `mode` from the same block although process() never puts them there — pass `scale`.  `half_index`: keep the
    reference's index into the accumulated robustness (it addresses the [H, W] map at half resolution)."""
    return _frame_count_denoise(image, r_acc, config, 0, config.radius_max, scale, half_index, mode)


def frame_count_denoising_gauss(image, r_acc, config, scale=None, half_index=True, mode=None):
    """Gaussian blur whose sigma grows where few frames were merged (utils_image.py:174-234).  Upstream this kernel does
    not compile (range() of the float 3 sigma); the build's window is |i|, |j| <= ceil(3 sigma)."""
    return _frame_count_denoise(image, r_acc, config, 1, config.sigma_max, scale, half_index, mode)

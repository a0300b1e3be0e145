"""Alg. 6-9 robustness (reference robustness.py)."""
import numpy as np
import torch

from . import _lib


import os as _os

_NO_GROUP = bool(_os.environ.get("HHSR_ROB_NO_GROUP"))  # the library's A/B switch, read once like the library reads it


def _wb3(white_balance):
    wb = [float(v) for v in (white_balance.tolist() if hasattr(white_balance, "tolist") else white_balance)]
    if len(wb) < 3:
        raise ValueError("white balance needs at least 3 gains")
    return wb[:3]


def compute_local_stats_from_raw(raw_img, cfa_pattern, white_balance, want_vars=True):
    """Guide image (Alg. 7, robustness.py:207-225) and its 3x3 local mean / variance (Alg. 8,
    :269-294) at guide resolution in one pass over the raw frame.  Returns (means, vars) [3, H/2, W/2]
    (vars = None when not wanted: comp frames only use their means)."""
    raw_img = _lib.f32c(raw_img)
    H, W = raw_img.shape
    if H % 2 or W % 2:
        raise ValueError(f"bayer frames need even dimensions, got {(H, W)}")
    means = torch.empty((3, H // 2, W // 2), dtype=torch.float32, device=raw_img.device)
    vars_ = torch.empty_like(means) if want_vars else None
    _lib.call("hhsr_rob_stats", _lib.ptr(raw_img), H, W, W, _lib.cfa_bytes(cfa_pattern), _lib.doubles(_wb3(white_balance)),
              _lib.ptr(means), _lib.ptr(vars_), _lib.stream())
    return means, vars_


def upscale_warp_stats(local_stats, tile_size=None, flow=None):
    """Dodgson-quadratic x2 upsampling of a [3, h, w] map, optionally warped by the per-tile flow
    (robustness.py:296-421).  +inf where the guide position falls outside the map."""
    local_stats = _lib.f32c(local_stats)
    nc, lh, lw = local_stats.shape
    if nc == 1:
        # `mode: grey`: the map keeps its size while the kernel keeps its hard-coded s = 2 (robustness.py:337-343, 358):
        # the top-left quadrant stretched over the frame, for the reference and the warped frames alike (reproduced)
        out = torch.empty_like(local_stats)
        ny, nx = (0, 0) if flow is None else flow.shape[:2]
        _lib.call("hhsr_mono_rob_upscale", _lib.ptr(local_stats), lh, lw, _lib.ptr(flow), ny, nx,
                  0 if flow is None else int(tile_size), _lib.ptr(out), _lib.stream())
        return out
    if nc != 3:
        raise ValueError("Incoherent number of channel : {}".format(nc))
    out = torch.empty((3, 2 * lh, 2 * lw), dtype=torch.float32, device=local_stats.device)
    if flow is None:
        _lib.call("hhsr_rob_upscale", _lib.ptr(local_stats), lh, lw, _lib.ptr(None), 0, 0, 0, _lib.ptr(out), _lib.stream())
    else:
        ny, nx, _ = flow.shape
        _lib.call("hhsr_rob_upscale", _lib.ptr(local_stats), lh, lw, _lib.ptr(flow), ny, nx, int(tile_size),
                  _lib.ptr(out), _lib.stream())
    return out


def init_robustness(ref_img, cfa_pattern, white_balance, config):
    """Reference-frame local means / variances at raw resolution (robustness.py:23-76)."""
    if not config.robustness.enabled:
        return None, None
    if config.mode != "bayer":  # the frame is its own one-channel guide image (robustness.py:62-66)
        from .kernels import mono_frame_stats

        m, v, _ = mono_frame_stats(ref_img, config, covs=False, want_vars=True)
        return upscale_warp_stats(m), upscale_warp_stats(v)
    m, v = compute_local_stats_from_raw(ref_img, cfa_pattern, white_balance)
    return upscale_warp_stats(m), upscale_warp_stats(v)


def _check_flow_view(flows, flow_rows):
    """`flow_rows` != (0, 0) makes the kernels read `before` / `after` tile rows OUTSIDE the tensor they are handed: that
    is only memory-safe for a true row-slice VIEW of a larger float32 field (a silent copy — dtype, device or contiguity
    conversion — would be read out of bounds)."""
    before, after = int(flow_rows[0]), int(flow_rows[1])
    if before == 0 and after == 0:
        return
    ny, nx, two = flows.shape
    if not (flows.is_cuda and flows.dtype == torch.float32 and flows.is_contiguous() and two == 2):
        raise ValueError("flow_rows needs a contiguous float32 device VIEW of the full flow field")
    off = flows.storage_offset()
    total = flows.untyped_storage().nbytes() // 4
    if before < 0 or after < 0 or off < before * nx * 2 or off + (ny + after) * nx * 2 > total:
        raise ValueError(f"flow_rows {tuple(flow_rows)}: the tile rows around the view are not inside its storage")


def compute_s(flows, M_th, s1, s2, flow_rows=(0, 0)):
    """Per-tile flow-irregularity weight (robustness.py:530-612).  `flow_rows` = (before, after): `flows` is a row
    slice (a VIEW) of a larger field with that many tile rows around it in memory — the multi-GPU row slabs — and the
    3 x 3 tile neighbourhood reads them, so the slice's first and last rows get the full field's weights."""
    ny, nx, _ = flows.shape
    _check_flow_view(flows, flow_rows)
    S = torch.empty((ny, nx), dtype=torch.float32, device=flows.device)
    _lib.call("hhsr_rob_s", _lib.ptr(flows), ny, nx, float(M_th), float(s1), float(s2), _lib.ptr(S),
              int(flow_rows[0]), int(flow_rows[1]), _lib.stream())
    return S


class RobustnessSum:
    """The accumulated robustness the way the reference keeps it when something DECIDES on it: a float64 sum of the
    per-frame maps (super_resolution.py:116-117, utils.py:93-120).  The accumulated-robustness denoiser compares the sum with
    `max_frame_count` (merge.py:223-228: widen the reference splat / overwrite instead of add), and a sum such as
    1 + 1 + 0.99999994 is 3 in float32 and not in float64 (found by the round-5 sweep: case 4300.15, 6 output values off by
    0.038).  The kernels keep reading a float32 map: `for_decisions()` hands them one whose `<=` / `<` relations to the
    threshold are those of the float64 sum.  (Where the sum is only reported — `robustness.save_mask` without the denoiser
    — it stays the float32 sum of the fused kernels)"""

    def __init__(self, shape, device):
        self.sum = torch.zeros(tuple(shape), dtype=torch.float64, device=device)

    def add(self, r):
        self.sum.add_(r)  # float32 -> float64: exact
        self._n = getattr(self, "_n", 0) + 1
        return self

    def add_many(self, rs, unfiltered=False):
        """The maps of several frames in ONE pass (hhsr_rob_sum: the float64 sum in frame order, like add() frame by
        frame — bit-identical —, without a read-modify-write of the float64 map per frame).
        unfiltered=True: `rs` are the thresholded maps R BEFORE the 5x5 local minimum (the fused merge applies it itself:
        merge_burst(local_min=True)); the minimum is taken on the way into the sum (HHSR_ROB_SUM_MIN5) = local_min() per
        frame, then add(), without the filtered maps ever being written."""
        rs = [_lib.f32c(r) for r in rs]
        H, W = self.sum.shape
        for i in range(0, len(rs), _lib.MAX_FRAMES):
            chunk = rs[i:i + _lib.MAX_FRAMES]
            flags = (1 if getattr(self, "_n", 0) else 0) | (2 if unfiltered else 0)  # HHSR_ROB_SUM_LOAD | _MIN5
            _lib.call("hhsr_rob_sum", _lib.ptr_array(chunk), len(chunk), int(H), int(W), flags, 0.0,
                      _lib.ptr(self.sum), None, None, _lib.stream(self.sum.device))
            self._n = getattr(self, "_n", 0) + len(chunk)
        return self

    def mask(self, rows=None):
        """The sum as the float32 map the API returns (`debug_dict['accumulated robustness']`)."""
        t = self.sum if rows is None else self.sum[rows[0]:rows[1]]
        return t.to(torch.float32)

    @staticmethod
    def decisions_of(sum64, max_frame_count):
        """float32 map a with  a <= mfc  <=>  sum <= mfc  and  a < mfc  <=>  sum < mfc  (the kernel compares (double) a).
        Exact for every sum and every threshold float32 can hold (the reference's are frame counts); for a threshold it
        cannot hold, a sum EQUAL to it (to the last float64 bit) lands on the side its float32 rounding lies on."""
        mfc = float(max_frame_count)
        m32 = np.float32(mfc)
        below = m32 if float(m32) < mfc else np.nextafter(m32, np.float32(-np.inf))  # largest float32 below the threshold
        above = m32 if float(m32) > mfc else np.nextafter(m32, np.float32(np.inf))   # smallest float32 above it
        a = sum64.to(torch.float32)
        a = torch.where((sum64 < mfc) & (a >= float(m32)), torch.full_like(a, float(below)), a)
        a = torch.where((sum64 > mfc) & (a <= float(m32)), torch.full_like(a, float(above)), a)
        return a

    def for_decisions(self, max_frame_count, rows=None):
        t = self.sum if rows is None else self.sum[rows[0]:rows[1]]
        if t.is_cuda and t.is_contiguous():  # one kernel instead of five elementwise passes (= decisions_of, tested)
            out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
            _lib.call("hhsr_rob_sum", _lib.ptr_array([]), 0, int(t.shape[0]), int(t.shape[1]), 1, float(max_frame_count),
                      _lib.ptr(t), None, _lib.ptr(out), _lib.stream(t.device))
            return out
        return self.decisions_of(t, max_frame_count)


def local_min(R, accumulate_into=None):
    """Alg. 9: 5x5 clamp-border minimum (robustness.py:641-686).  `accumulate_into` (float32 [H, W]) gets
    += r in the same pass (the reference's separate add() of super_resolution.py:158-159)."""
    r = torch.empty_like(R)
    _lib.call("hhsr_local_min5", _lib.ptr(R), R.shape[0], R.shape[1], _lib.ptr(r), _lib.ptr(accumulate_into),
              _lib.stream())
    return r


_curves_cache = {}  # (device, content) -> device copies: a serving loop uploads a burst's curves once


def noise_curves_to_device(std_curve, diff_curve, device):
    """float64 device copies of the two noise curves (super_resolution.py:98-99), cached by content: repeated bursts
    with the same curves do no host-to-device copy (which also keeps main() capturable in a HIP graph)."""
    s = np.ascontiguousarray(np.asarray(std_curve, dtype=np.float64))
    d = np.ascontiguousarray(np.asarray(diff_curve, dtype=np.float64))
    key = (str(device), s.tobytes(), d.tobytes())
    hit = _curves_cache.get(key)
    if hit is None:
        if len(_curves_cache) >= 8:
            _curves_cache.pop(next(iter(_curves_cache)))
        hit = _curves_cache[key] = (torch.as_tensor(s, device=device).contiguous(),
                                    torch.as_tensor(d, device=device).contiguous())
    return hit


def noise_sigma_sq(ref_local_means, ref_local_stds, std_curve):
    """sigma^2 = sum_c max(var_c, sigma_t(mu_c)^2) of the reference frame (robustness.py:505-528): it does
    not depend on the compared frame, so a burst computes it once.
    Returns (sigma_sq float32 [H, W], curve_index int32 [H, W] or None): the second plane holds the three
    noise-curve indices round(1000 mu_c) of every pixel (10 bits each), equally frame-independent."""
    _, H, W = ref_local_means.shape
    out = torch.empty((H, W), dtype=torch.float32, device=ref_local_means.device)
    idx = torch.empty((H, W), dtype=torch.int32, device=out.device) if std_curve.numel() <= 1024 else None
    _lib.call("hhsr_rob_sigma", _lib.ptr(ref_local_means), _lib.ptr(ref_local_stds), H, W, _lib.ptr(std_curve),
              int(std_curve.numel()), _lib.ptr(out), _lib.ptr(idx), _lib.stream())
    return out, idx


def ref_planes(guide_means, guide_vars, std_curve):
    """Reference-frame state of the robustness in one pass (once per burst): the upsampled local means
    [3, H, W] (init_robustness, robustness.py:23-76) and noise_sigma_sq()'s (sigma_sq, curve_index) — bit-identical
    to upscale_warp_stats() x 2 + noise_sigma_sq(), without materialising the upsampled variances."""
    guide_means, guide_vars = _lib.f32c(guide_means), _lib.f32c(guide_vars)
    _, lh, lw = guide_means.shape
    dev = guide_means.device
    means = torch.empty((3, 2 * lh, 2 * lw), dtype=torch.float32, device=dev)
    sig = torch.empty((2 * lh, 2 * lw), dtype=torch.float32, device=dev)
    idx = torch.empty((2 * lh, 2 * lw), dtype=torch.int32, device=dev) if std_curve.numel() <= 1024 else None
    _lib.call("hhsr_ref_planes", _lib.ptr(guide_means), _lib.ptr(guide_vars), lh, lw, _lib.ptr(std_curve),
              int(std_curve.numel()), _lib.ptr(means), _lib.ptr(sig), _lib.ptr(idx), _lib.stream())
    return means, (sig, idx)


def compute_robustness(comp_img, ref_local_means, ref_local_stds, flows, cfa_pattern, white_balance, noise_model,
                       config, return_R=False, accumulate_into=None, ref_sigma_sq=None, comp_means=None,
                       fuse_local_min=False, flow_rows=(0, 0)):
    """Alg. 6 (robustness.py:79-170): r float32 [H, W].  3 kernels instead of the reference's 8:
    guide + local stats; fused warp-upsample / colour distance / noise model / threshold; 5x5 min."""
    comp_img = _lib.f32c(comp_img)
    if not config.robustness.enabled:
        ones = torch.ones_like(comp_img)
        if accumulate_into is not None:
            accumulate_into += ones
        return ones
    ts = config.block_matching.tuning.tile_size
    t = config.robustness.tuning
    std_curve, diff_curve = noise_model
    assert std_curve.dtype == torch.float64 and diff_curve.dtype == torch.float64 and std_curve.is_cuda
    H, W = comp_img.shape
    ny, nx, _ = flows.shape
    if config.mode != "bayer":
        return _compute_robustness_mono(comp_img, ref_local_means, ref_local_stds, flows, noise_model, config, return_R,
                                        accumulate_into, ref_sigma_sq, comp_means, fuse_local_min)
    if ref_sigma_sq is None:  # a BurstPipeline passes the per-burst map; stand-alone callers get it here
        ref_sigma_sq = noise_sigma_sq(ref_local_means, ref_local_stds, std_curve)
    cm = comp_means  # a BurstPipeline gets them from the fused per-frame pass (kernels.frame_stats)
    if cm is None:
        cm, _ = compute_local_stats_from_raw(comp_img, cfa_pattern, white_balance, want_vars=False)
    S = compute_s(flows, t.Mt, t.s1, t.s2, flow_rows)
    R = torch.empty((H, W), dtype=torch.float32, device=comp_img.device)
    sigma_sq, curve_index = ref_sigma_sq
    _lib.call("hhsr_rob_frame", _lib.ptr(cm), H // 2, W // 2, _lib.ptr(ref_local_means), _lib.ptr(sigma_sq),
              _lib.ptr(curve_index), _lib.ptr(flows), ny, nx, int(ts), _lib.ptr(S), _lib.ptr(diff_curve), int(diff_curve.numel()),
              float(t.t), _lib.ptr(R), _lib.stream())
    if fuse_local_min:  # the caller's merge applies Alg. 9 itself (merge_burst(..., local_min=True))
        assert accumulate_into is None and not return_R
        return R
    r = local_min(R, accumulate_into)
    return (r, R) if return_R else r


def mono_sigma_sq(ref_local_means, ref_local_stds, std_curve):
    """noise_sigma_sq() of a monochrome burst: sigma^2 float32 [H, W] (robustness.py:505-528 with one channel)."""
    _, H, W = ref_local_means.shape
    out = torch.empty((H, W), dtype=torch.float32, device=ref_local_means.device)
    _lib.call("hhsr_mono_rob_sigma", _lib.ptr(ref_local_means), _lib.ptr(ref_local_stds), H, W, _lib.ptr(std_curve),
              int(std_curve.numel()), _lib.ptr(out), _lib.stream())
    return out


def _compute_robustness_mono(comp_img, ref_local_means, ref_local_stds, flows, noise_model, config, return_R,
                             accumulate_into, ref_sigma_sq, comp_means, fuse_local_min=False):
    """`mode: grey` (robustness.py:79-170 with the one-channel branches): 3x3 means of the frame itself, fused
    warp / distance / noise model / threshold, 5x5 minimum."""
    from .kernels import mono_frame_stats

    ts = config.block_matching.tuning.tile_size
    t = config.robustness.tuning
    std_curve, diff_curve = noise_model
    H, W = comp_img.shape
    ny, nx, _ = flows.shape
    sigma_sq = ref_sigma_sq if ref_sigma_sq is not None else mono_sigma_sq(ref_local_means, ref_local_stds, std_curve)
    cm = comp_means if comp_means is not None else mono_frame_stats(comp_img, config, covs=False)[0]
    S = compute_s(flows, t.Mt, t.s1, t.s2)
    R = torch.empty((H, W), dtype=torch.float32, device=comp_img.device)
    _lib.call("hhsr_mono_rob_frame", _lib.ptr(cm), H, W, _lib.ptr(ref_local_means), _lib.ptr(sigma_sq), _lib.ptr(flows),
              ny, nx, int(ts), _lib.ptr(S), _lib.ptr(diff_curve), int(diff_curve.numel()), float(t.t), _lib.ptr(R),
              _lib.stream())
    if fuse_local_min:  # the caller's merge applies Alg. 9 itself (merge_burst(..., local_min=True))
        assert accumulate_into is None and not return_R
        return R
    r = local_min(R, accumulate_into)
    return (r, R) if return_R else r


def compute_robustness_group(comp_imgs, ref_local_means, flows, noise_model, config, ref_sigma_sq, comp_means,
                             accumulate_into=None, fuse_local_min=False, flow_rows=(0, 0)):
    """compute_robustness() for several frames of one burst: the fused kernel runs once per 4 frames and reads the
    reference-frame planes (20 of the 27 bytes per pixel and frame) once per group (hhsr_rob_frames); per frame the
    result is bit-identical to compute_robustness().  Needs the per-burst (sigma_sq, curve_index) and the frames' guide
    means (kernels.frame_stats).  Returns the list of r (or of the thresholded maps R with fuse_local_min)."""
    assert config.robustness.enabled and config.mode == "bayer"
    ts = config.block_matching.tuning.tile_size
    t = config.robustness.tuning
    _, diff_curve = noise_model
    H, W = comp_imgs[0].shape
    ny, nx, _ = flows[0].shape
    sigma_sq, curve_index = ref_sigma_sq
    # the grouped kernel evaluates the per-tile flow-irregularity weight S itself (one launch less per frame); the
    # per-frame fall-back kernels of hhsr_rob_frames need the S maps — same test as in the library
    inline_s = (int(ts) % 16 == 0 and W % 4 == 0 and curve_index is not None and diff_curve.numel() <= 1024
                and not _NO_GROUP)
    for f in flows:
        _check_flow_view(f, flow_rows)
    S = None if inline_s else [compute_s(f, t.Mt, t.s1, t.s2, flow_rows) for f in flows]
    R = [torch.empty((H, W), dtype=torch.float32, device=comp_imgs[0].device) for _ in flows]
    _lib.call("hhsr_rob_frames", _lib.ptr_array(comp_means), len(flows), H // 2, W // 2, _lib.ptr(ref_local_means),
              _lib.ptr(sigma_sq), _lib.ptr(curve_index), _lib.ptr_array(flows), ny, nx, int(ts),
              None if S is None else _lib.ptr_array(S), float(t.Mt), float(t.s1), float(t.s2),
              _lib.ptr(diff_curve), int(diff_curve.numel()), float(t.t), _lib.ptr_array(R), int(flow_rows[0]),
              int(flow_rows[1]), _lib.stream())
    if fuse_local_min:
        assert accumulate_into is None
        return R
    return [local_min(r_, accumulate_into) for r_ in R]

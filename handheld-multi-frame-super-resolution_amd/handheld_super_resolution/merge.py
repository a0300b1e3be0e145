"""Alg. 4 / Alg. 11 accumulation (reference merge.py) plus the fused burst merge."""
import os

import torch

from . import _lib


# kflags of the C ABI (include/hhsr.h)
KERNEL_ISO, WEIGHT_F64, FORCE_GENERIC, FORCE_TILE, FORCE_X2V1, SENSOR_MONO = 1, 2, 4, 8, 16, 32
REF_DIVIDE, REF_FAST = 64, 128  # hhsr_accumulate_ref only (include/hhsr.h HHSR_REF_DIVIDE, HHSR_REF_FAST)
_FORCE = {"auto": 0, "generic": FORCE_GENERIC, "tile": FORCE_TILE, "x2_v1": FORCE_X2V1}
# process-wide A/B switches, read once like the library reads them
_ENV_FORCE = ((FORCE_GENERIC if os.environ.get("HHSR_MERGE_NO_LDS") else 0) |
              (FORCE_TILE if os.environ.get("HHSR_MERGE_NO_QUAD") else 0) |
              (FORCE_X2V1 if os.environ.get("HHSR_MERGE_X2_V1") else 0))


def _common(config):
    """(scale, kflags): bit 0 = iso kernel, bit 1 = float64 weight chain (config.hip.weight_fp64, the
    reference's Numba typing; default float32 weights with float64 geometry), bits 2-4 = restriction of
    hhsr_merge_burst's kernel choice (config.hip.merge_kernel: auto | generic | tile | x2_v1; validation only),
    bit 5 = `mode: grey` (one channel, per-pixel covariances)."""
    hip = config.get("hip", None) if hasattr(config, "get") else None
    f64 = bool(hip.get("weight_fp64", False)) if hip is not None else False
    force = _FORCE[str(hip.get("merge_kernel", "auto"))] if hip is not None else 0
    return float(config.scale), ((KERNEL_ISO if config.merging.kernel == "iso" else 0) | (WEIGHT_F64 if f64 else 0) |
                                 force | _ENV_FORCE | (SENSOR_MONO if config.mode != "bayer" else 0))


def merge(comp_img, alignments, covs, r, num, den, cfa_pattern, config):
    """Accumulate one non-reference frame into num / den in place (merge.py:236-288)."""
    scale, kflags = _common(config)
    ts = config.block_matching.tuning.tile_size
    H, W = comp_img.shape
    ny, nx, _ = alignments.shape
    sH, sW, _ = num.shape
    _lib.call("hhsr_accumulate", _lib.ptr(comp_img), H, W, W, _lib.ptr(alignments), ny, nx, int(ts),
              _lib.ptr(covs), _lib.ptr(r), _lib.cfa_bytes(cfa_pattern), scale, kflags, _lib.ptr(num), _lib.ptr(den),
              sH, sW, _lib.stream())


def merge_ref(ref_img, kernels, num, den, cfa_pattern, config, acc_rob=None, divide=False, fast=False):
    """Accumulate the reference frame (merge.py:22-80); with the accumulated-robustness denoiser enabled
    the window widens / the pixel is overwritten where few frames were merged.
    divide=True (not in the reference's signature): the normalisation that follows in main() (super_resolution.py:187-190)
    in the same pass — num = (num (+) ref) / (den (+) ref), `den` is left as it was; bit-identical to merge_ref + divide.
    fast=True: float32 weights (the fused merge's reference-frame arithmetic) for the pixels the denoiser does not widen,
    outside the border bands; the default is the reference's float64 chain on every pixel (<= 2e-5 relative apart)."""
    scale, kflags = _common(config)
    if divide:
        kflags |= REF_DIVIDE
    if fast and not (kflags & WEIGHT_F64):
        kflags |= REF_FAST
    H, W = ref_img.shape
    sH, sW, _ = num.shape
    den_cfg = config.accumulated_robustness_denoiser
    if den_cfg.enabled:
        if acc_rob is None:
            raise ValueError("accumulated robustness denoiser enabled but no accumulated robustness given")
        acc = _lib.f32c(acc_rob)
        rad_max, mult, mfc = int(den_cfg.merge.rad_max), float(den_cfg.merge.max_multiplier), float(den_cfg.merge.max_frame_count)
    else:
        acc, rad_max, mult, mfc = None, 0, 0.0, 0.0
    _lib.call("hhsr_accumulate_ref", _lib.ptr(ref_img), H, W, W, _lib.ptr(kernels), _lib.cfa_bytes(cfa_pattern),
              scale, kflags, _lib.ptr(acc), rad_max, mult, mfc, _lib.ptr(num), _lib.ptr(den), sH, sW, _lib.stream())


def can_fuse_acc_r(config):
    """merge_burst can also produce the accumulated robustness (integer scales)."""
    return float(config.scale).is_integer()


def can_fuse_local_min(config, shape):
    """merge_burst can take the thresholded maps R and apply the 5x5 local minimum itself (the wave-per-parity-class
    kernels: scale 2, or scale 3 with W % 4 == 0; tile size a multiple of 16; float32 weights) — mirrors the test in
    hhsr_merge_burst."""
    scale, kflags = _common(config)
    H, W = shape
    ok = not (kflags & (WEIGHT_F64 | FORCE_GENERIC | FORCE_TILE)) and int(config.block_matching.tuning.tile_size) % 16 == 0
    if kflags & SENSOR_MONO:  # monochrome: the x2 tile kernel only
        return ok and scale == 2.0 and not (kflags & FORCE_X2V1)
    return ok and ((scale == 2.0 and H % 2 == 0 and W % 2 == 0) or (scale == 3.0 and W % 4 == 0 and _is_bayer(config)))


def merge_burst(frames, ref_img, ref_kernels, num, den, cfa_pattern, config, load_acc=False, do_ref=True,
                divide=True, store_den=False, acc_r=None, rows=None, out_height=None, local_min=False, lr_row_offset=0):
    """Fused merge of a whole (shard of a) burst: `frames` is a list of (raw, flow, covs, r).  Per output
    pixel the frames are summed in list order with the accumulators in registers — the same float32
    order as successive merge() calls — then the reference frame is added and the result normalised,
    writing `num` once.  Not usable with the accumulated-robustness denoiser (its
    overwrite rule needs the sequential merge_ref).  `acc_r` (float32 [H, W], integer scales) receives the sum
    of the frames' robustness maps in the same pass.  `rows = (row0, nrows)` restricts the launch to a slab of
    output rows; `num` / `den` are then [nrows, sW, 3] slabs and `out_height` the full output height.
    `local_min`: the frames carry the thresholded maps R (compute_robustness(..., fuse_local_min=True)) and the
    5x5 minimum of Alg. 9 is taken inside the merge (see can_fuse_local_min).
    `lr_row_offset`: the frames are sub-images starting at this raw row of the full frame (multi-GPU row slabs):
    positions are evaluated in full-frame coordinates (see include/hhsr.h)."""
    scale, kflags = _common(config)
    if do_ref and config.accumulated_robustness_denoiser.enabled:
        raise ValueError("merge_burst cannot apply the accumulated robustness denoiser; use merge_ref")
    ts = config.block_matching.tuning.tile_size
    sH, sW, _ = num.shape
    row0, nrows = (0, sH) if rows is None else rows
    if rows is not None:
        assert num.shape[0] == nrows and out_height is not None
        sH = int(out_height)
    flags = (1 if load_acc else 0) | (2 if do_ref else 0) | (4 if divide else 0) | (8 if store_den else 0)
    if local_min and frames:
        flags |= 16
    if frames:
        H, W = frames[0][0].shape
        ny, nx, _ = frames[0][1].shape
    else:
        H, W = ref_img.shape
        ny = nx = 0
    cfa = _lib.cfa_bytes(cfa_pattern)
    chunks = [frames[i:i + _lib.MAX_FRAMES] for i in range(0, len(frames), _lib.MAX_FRAMES)] or [[]]
    if len(chunks) > 1 and den is None:  # bursts longer than one launch holds chain through num / den
        den = torch.empty_like(num)
    for ci, chunk in enumerate(chunks):
        last = ci == len(chunks) - 1
        f = flags
        if ci > 0:
            f |= 1
        if not last:  # intermediate chunk: keep raw sums
            f = (f & ~(2 | 4)) | 8
        _lib.call("hhsr_merge_burst", _lib.ptr_array([c[0] for c in chunk]), _lib.ptr_array([c[1] for c in chunk]),
                  _lib.ptr_array([c[2] for c in chunk]), _lib.ptr_array([c[3] for c in chunk]), len(chunk),
                  H, W, W, ny, nx, int(ts), _lib.ptr(ref_img if (f & 2) else None),
                  _lib.ptr(ref_kernels if (f & 2) else None), cfa, scale, kflags, f, _lib.ptr(num), _lib.ptr(den),
                  _lib.ptr(acc_r if chunk else None), sH, sW, int(row0), int(nrows), int(lr_row_offset), _lib.stream())


def _is_bayer(config):
    """2 x 2 colour layout with red and blue on one diagonal and green on the other (what the wave-per-class kernels
    fold their parity classes into; other layouts take the first-generation tile kernels) — mirrors cfa_is_bayer()."""
    try:
        c = [int(v) for row in config.exif.cfa_pattern for v in row]
    except Exception:
        return False
    for k in range(4):
        if c[k] == 0:
            return c[3 - k] == 2 and c[k ^ 1] == 1 and c[k ^ 2] == 1
    return False


def can_chain(config, shape):
    """merge_burst_chain applies: the wave-per-class x2 kernel (same conditions as can_fuse_local_min at scale 2, Bayer)."""
    scale, kflags = _common(config)
    return scale == 2.0 and not (kflags & SENSOR_MONO) and can_fuse_local_min(config, shape) and _is_bayer(config)


def chain_buffer(shape, device):
    """Parking space of the parity-class accumulators between the two launches of merge_burst_chain (1.6 GB at 12 MP)."""
    H, W = shape
    return torch.empty((_lib.load().hhsr_merge_chain_bytes(int(H), int(W)) // 4,), dtype=torch.float32, device=device)


def merge_burst_chain(frames, n_done, ref_img, ref_kernels, num, cfa_pattern, config, class_acc, last, acc_r=None,
                      local_min=False):
    """One link of the fused x2 merge as a chain of launches, bit-identical to merge_burst(all frames, ..., do_ref=True,
    divide=True) (include/hhsr.h, hhsr_merge_burst_chain).  `frames`: the frames that have arrived so far, of which the
    first `n_done` were merged by earlier links; this link adds frames[n_done:] to the parity-class accumulators parked in
    `class_acc` (chain_buffer).  `last`: also add the reference frame, normalise and write `num` (and `acc_r`) — the
    earlier links can run while the remaining frames are still on their way."""
    scale, kflags = _common(config)
    assert 0 <= n_done <= len(frames) <= _lib.MAX_FRAMES and (n_done < len(frames) or last)
    H, W = frames[0][0].shape
    ny, nx, _ = frames[0][1].shape
    sH, sW, _ = num.shape
    flags = (_lib.MERGE_LOAD_CLASSES if n_done else 0) | \
            ((_lib.MERGE_DO_REF | _lib.MERGE_DIVIDE) if last else _lib.MERGE_STORE_CLASSES)
    if local_min:
        flags |= _lib.MERGE_LOCAL_MIN
    if last and not n_done:
        raise ValueError("a chain of one link is merge_burst()")
    ts = config.block_matching.tuning.tile_size
    _lib.call("hhsr_merge_burst_chain", _lib.ptr_array([c[0] for c in frames]), _lib.ptr_array([c[1] for c in frames]),
              _lib.ptr_array([c[2] for c in frames]), _lib.ptr_array([c[3] for c in frames]), len(frames), H, W, W, ny, nx,
              int(ts), _lib.ptr(ref_img if last else None), _lib.ptr(ref_kernels if last else None),
              _lib.cfa_bytes(cfa_pattern), scale, kflags, flags, _lib.ptr(num), _lib.ptr(None),
              _lib.ptr(acc_r if last else None), sH, sW, _lib.ptr(class_acc), int(n_done), _lib.stream())

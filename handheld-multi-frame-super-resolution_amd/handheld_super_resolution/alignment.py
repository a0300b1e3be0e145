"""Coarse-to-fine alignment driver (reference alignment.py).

Flow convention: alignments[ty, tx] = (dx, dy), moving(p + flow) ~= ref(p).  Pyramids are COARSE FIRST
like the reference's lists."""
import torch
import torch.nn.functional as F

from . import _lib
from .ICA import init_ica, align_lvl_ica
from .block_matching import align_lvl_block_matching_L2, align_lvl_block_matching_L1
from .utils_image import cuda_downsample, cuda_downsample_batch


def build_gaussian_pyramid(image, factors=[1, 2, 4, 4], kernel="gaussian"):
    """alignment.py:74-82: compact contiguous levels, coarse first."""
    pyramid = [cuda_downsample(image, kernel, factors[0])]
    for factor in factors[1:]:
        pyramid.append(cuda_downsample(pyramid[-1], kernel, factor))
    pyramid = [lvl.reshape(lvl.shape[-2:]) for lvl in pyramid]
    return pyramid[::-1]


def build_gaussian_pyramids(images, factors=[1, 2, 4, 4]):
    """build_gaussian_pyramid() of several frames of one shape, one launch per level (coarse first per frame)."""
    levels = [cuda_downsample_batch(images, factors[0])]
    for factor in factors[1:]:
        levels.append(cuda_downsample_batch(levels[-1], factor))
    return [[lvl[i] for lvl in levels[::-1]] for i in range(len(images))]


def init_alignment(ref_img, config):
    """Reference-side precompute (alignment.py:20-72): circular pad to a multiple of the tile size,
    pyramid, per-level gradients + Hessian.

    Returns the reference's 6-tuple (pyramid, tiled_pyr, tiled_fft, gradx, grady, hessian), coarse first.
    The tiled / FFT'd copies of the reference tiles exist upstream only to feed the FFT correlation; the
    LDS block-matching kernel reads the level directly, so those two lists hold None."""
    ref_img = _lib.f32c(ref_img)
    h, w = ref_img.shape
    bm = config.block_matching.tuning
    Ts, tss, factors = bm.tile_size, bm.tile_sizes, bm.factors
    pb = (Ts - h % Ts) * (h % Ts != 0)
    pr = (Ts - w % Ts) * (w % Ts != 0)
    if pb or pr:
        padded = torch.empty((h + pb, w + pr), dtype=torch.float32, device=ref_img.device)
        _lib.call("hhsr_pad_circular", _lib.ptr(ref_img), h, w, w, _lib.ptr(padded), h + pb, w + pr, w + pr,
                  _lib.stream())
    else:
        padded = ref_img
    pyramid = build_gaussian_pyramid(padded, factors)
    gxs, gys, hs = [], [], []
    for i, lvl in enumerate(pyramid):
        ts = tss[len(factors) - i - 1]
        if lvl.shape[0] // ts < 1 or lvl.shape[1] // ts < 1:
            raise ValueError(f"pyramid level of shape {tuple(lvl.shape)} cannot be divided into tiles of size {ts}")
        gx, gy, hess = init_ica(lvl, ts, config)
        gxs.append(gx)
        gys.append(gy)
        hs.append(hess)
    none = [None] * len(pyramid)
    return pyramid, list(none), list(none), gxs, gys, hs


def upscale_lvl(alignments, npatchs, l, config):
    """Re-tile and scale the flow for the next finer level (alignment.py:150-172)."""
    bm = config.block_matching.tuning
    new_ts, prev_ts = bm.tile_sizes[l], bm.tile_sizes[l + 1]
    up = bm.factors[l + 1]
    rep = up // (new_ts // prev_ts)
    mode = bm.flow_upscale_mode
    sny, snx, _ = alignments.shape
    if mode == "nearest":
        out = torch.empty((npatchs[0], npatchs[1], 2), dtype=torch.float32, device=alignments.device)
        _lib.call("hhsr_flow_upscale_nearest", _lib.ptr(alignments), sny, snx, _lib.ptr(out), npatchs[0], npatchs[1],
                  rep, float(up), _lib.stream())
        return out
    # bilinear / bicubic: the reference itself delegates to F.interpolate on this <= 188x250x2 field
    ups = F.interpolate(alignments.permute(2, 0, 1)[None], scale_factor=rep, mode=mode)[0].permute(1, 2, 0)
    ups = ups * up
    py, px = npatchs[0] - ups.shape[0], npatchs[1] - ups.shape[1]
    if py > 0 or px > 0:
        ups = F.pad(ups, (0, 0, 0, max(px, 0), 0, max(py, 0)), mode="constant", value=0)
    return ups[: npatchs[0], : npatchs[1]].contiguous()


def _fused_level(l, config):
    """(metric code, ts, r) when level l runs on the fused block-matching + ICA kernel, else None."""
    bm = config.block_matching.tuning
    ts, r = bm.tile_sizes[l], bm.search_radii[l]
    hip = config.get("hip", None) if hasattr(config, "get") else None
    fused = True if hip is None else bool(hip.get("fused_align", True))
    code = {"L2": 0, "L1": 1, "L1_ref_effective": 2}.get(bm.metrics[l])
    if fused and code is not None and ts in (8, 16, 32) and r in (1, 2, 4) and not (code != 0 and ts == 8):
        return code, ts, r
    return None


def align_lvl(ref_lvl, tyled_pyr_lvl, ref_fft_lvl, ref_gradx_lvl, ref_grady_lvl, ref_hessian_lvl, moving_lvl,
              alignments, l, config, coarse=None):
    """Block matching then ICA on one level (alignment.py:125-147).  For tiles up to 32 pixels both steps
    run in ONE fused kernel (hhsr_align_level; config.hip.fused_align: false selects the two-kernel path,
    which is also what 64-pixel tiles use).  `coarse` (fused kernel only): where the incoming flow comes from
    instead of `alignments` — "zero" or (coarser_flow, rep, mult) for the nearest-neighbour upscaling of
    upscale_lvl() fused into the launch; `alignments` then only receives the result."""
    metric = config.block_matching.tuning.metrics[l]
    fl = _fused_level(l, config)
    if fl is not None:
        code, ts, r = fl
        ny, nx, _ = alignments.shape
        mh, mw = moving_lvl.shape
        rh, rw = ref_lvl.shape
        assert ref_lvl.is_contiguous() and moving_lvl.is_contiguous() and alignments.is_contiguous()
        if coarse is None:
            cptr, cny, cnx, rep, mult = None, 0, 0, 0, 1.0
        elif isinstance(coarse, str):
            cptr, cny, cnx, rep, mult = None, 0, 0, -1, 1.0
        else:
            cf, rep, mult = coarse
            assert cf.is_contiguous() and cf.dtype == torch.float32
            cptr, (cny, cnx) = cf, cf.shape[:2]
        _lib.call("hhsr_align_level", _lib.ptr(ref_lvl), rh, rw, rw, _lib.ptr(ref_hessian_lvl), _lib.ptr(moving_lvl),
                  mh, mw, mw, _lib.ptr(alignments), ny, nx, ts, r, code, int(config.ica.tuning.n_iter),
                  _lib.ptr(cptr), int(cny), int(cnx), int(rep), float(mult), _lib.stream())
        return
    assert coarse is None, "fused flow upscaling needs the fused level kernel"
    if metric == "L2":
        align_lvl_block_matching_L2(ref_lvl, ref_fft_lvl, moving_lvl, alignments, l, config)
    elif metric == "L1":
        align_lvl_block_matching_L1(ref_lvl, moving_lvl, alignments, l, config)
    elif metric == "L1_ref_effective":
        align_lvl_block_matching_L1(ref_lvl, moving_lvl, alignments, l, config, effective=True)
    else:
        raise ValueError("Unknown block matching metric {}".format(metric))
    align_lvl_ica(ref_lvl, ref_gradx_lvl, ref_grady_lvl, ref_hessian_lvl, moving_lvl, alignments, l, config)


def align(ref_pyramid, tyled_pyr, ref_tiled_fft, ref_gradx, ref_grady, ref_hessian, img, config,
          moving_pyramid=None):
    """Coarse-to-fine alignment of one grey frame (alignment.py:84-123).  Everything is enqueued on
    torch's current stream: no host synchronisation between levels (the reference needs a
    cuda.synchronize() per level to order its torch and Numba streams).  `moving_pyramid`: the frame's
    pyramid when the caller already built it (it does not depend on the reference frame)."""
    img = _lib.f32c(img)
    factors = config.block_matching.tuning.factors
    if moving_pyramid is None:
        moving_pyramid = build_gaussian_pyramid(img, factors)
    alignments = None
    n = len(ref_pyramid)
    for i in range(n):
        l = n - i - 1
        ts = config.block_matching.tuning.tile_sizes[l]
        grid = (ref_pyramid[i].shape[0] // ts, ref_pyramid[i].shape[1] // ts)
        bm = config.block_matching.tuning
        coarse = None
        q = bm.tile_sizes[l] // bm.tile_sizes[l + 1] if alignments is not None else 1
        rep = bm.factors[l + 1] // q if (alignments is not None and q > 0) else 1
        if (_fused_level(l, config) is not None and rep > 0 and
                (alignments is None or bm.flow_upscale_mode == "nearest")):
            # the level kernel takes its incoming flow straight from the coarser level (or zero): no separate
            # upscaling / memset launch
            if alignments is None:
                coarse = "zero"
            else:
                coarse = (alignments, rep, float(bm.factors[l + 1]))
            alignments = torch.empty((*grid, 2), dtype=torch.float32, device=img.device)
        elif alignments is None:
            alignments = torch.zeros((*grid, 2), dtype=torch.float32, device=img.device)
        else:
            alignments = upscale_lvl(alignments, grid, l, config)
        align_lvl(ref_pyramid[i], tyled_pyr[i], ref_tiled_fft[i], ref_gradx[i], ref_grady[i], ref_hessian[i],
                  moving_pyramid[i], alignments, l, config, coarse=coarse)
    return alignments


def can_align_batch(config):
    """Every level runs on the fused level kernel with the coarser flow read in place (see align())."""
    bm = config.block_matching.tuning
    n = len(bm.factors)
    for l in range(n):
        if _fused_level(l, config) is None:
            return False
        if l + 1 < n:
            q = bm.tile_sizes[l] // bm.tile_sizes[l + 1]
            if q <= 0 or bm.factors[l + 1] // q <= 0 or bm.flow_upscale_mode != "nearest":
                return False
    return True


def align_batch(ref_pyramid, ref_hessian, moving_pyramids, config):
    """align() of several frames against one reference pyramid: ONE launch per level for the whole chunk
    (hhsr_align_level_batch) — the coarse levels are 7-27 us launches of a few hundred workgroups that cannot fill the
    GPU one frame at a time.  Needs can_align_batch(config); per frame bit-identical to align().
    Returns the list of finest-level flow fields (views of one [n, ny, nx, 2] tensor)."""
    bm = config.block_matching.tuning
    n_lvl, nf = len(ref_pyramid), len(moving_pyramids)
    dev = moving_pyramids[0][0].device
    prev = None
    for i in range(n_lvl):
        l = n_lvl - i - 1
        code, ts, r = _fused_level(l, config)
        ref_lvl = ref_pyramid[i]
        rh, rw = ref_lvl.shape
        ny, nx = rh // ts, rw // ts
        movs = [p[i] for p in moving_pyramids]
        mh, mw = movs[0].shape
        flows = list(torch.empty((nf, ny, nx, 2), dtype=torch.float32, device=dev).unbind(0))
        if prev is None:
            cptr, cny, cnx, rep, mult = None, 0, 0, -1, 1.0
        else:
            q = bm.tile_sizes[l] // bm.tile_sizes[l + 1]
            rep, mult = bm.factors[l + 1] // q, float(bm.factors[l + 1])
            cptr, (cny, cnx) = _lib.ptr_array(prev), prev[0].shape[:2]
        _lib.call("hhsr_align_level_batch", _lib.ptr(ref_lvl), rh, rw, rw, _lib.ptr(ref_hessian[i]), _lib.ptr_array(movs),
                  nf, mh, mw, mw, _lib.ptr_array(flows), ny, nx, ts, r, code, int(config.ica.tuning.n_iter), cptr,
                  int(cny), int(cnx), int(rep), float(mult), _lib.stream())
        prev = flows
    return prev

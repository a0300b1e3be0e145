"""Lucas-Kanade inverse-compositional refinement (reference ICA.py)."""
import torch

from . import _lib


def init_ica(image, tile_size, config=None):
    """Gradients (un-normalised [-1,0,1], zero border) and per-tile 2x2 Hessian (ICA.py:15-34),
    one fused HIP kernel.  Returns (gradx, grady, hessian[ny, nx, 2, 2])."""
    image = _lib.f32c(image)
    H, W = image.shape
    ny, nx = H // tile_size, W // tile_size
    gx = torch.empty_like(image)
    gy = torch.empty_like(image)
    hess = torch.empty((ny, nx, 2, 2), dtype=torch.float32, device=image.device)
    _lib.call("hhsr_grad_hessian", _lib.ptr(image), H, W, W, tile_size, _lib.ptr(gx), _lib.ptr(gy),
              _lib.ptr(hess), _lib.stream())
    return gx, gy, hess


def align_lvl_ica(ref_img, ref_gradx_lvl, ref_grady_lvl, ref_hessian_lvl, moving_lvl, alignment, l, config):
    """n_iter Gauss-Newton steps per tile, in place on `alignment` (ICA.py:78-103)."""
    ts = config.block_matching.tuning.tile_sizes[l]
    if ts not in (8, 16, 32, 64):
        raise NotImplementedError("ICA kernel for tile size {} not implemented".format(ts))
    ny, nx, _ = alignment.shape
    mh, mw = moving_lvl.shape
    compat = config.get("compat", None) if hasattr(config, "get") else None
    bug = 1 if (compat is None or compat.get("ica64_row_bug", True)) else 0
    assert ref_img.is_contiguous() and moving_lvl.is_contiguous() and alignment.is_contiguous()
    _lib.call("hhsr_ica", _lib.ptr(ref_img), _lib.ptr(ref_gradx_lvl), _lib.ptr(ref_grady_lvl), ref_img.shape[1],
              _lib.ptr(ref_hessian_lvl), _lib.ptr(moving_lvl), mh, mw, mw, _lib.ptr(alignment), ny, nx, ts,
              int(config.ica.tuning.n_iter), bug, _lib.stream())

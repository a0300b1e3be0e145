"""Oracle: coarse-to-fine alignment (restates reference alignment.py, block_matching.py, ICA.py).

Test infrastructure.  Flow convention everywhere: ``flow[ty, tx] = (dx, dy)`` with
moving(p + flow) ~= ref(p) (reference alignment.py:84-123).
"""
import numpy as np

from .pyramid import build_gaussian_pyramid

F32 = np.float32


def _rne(x):
    """round-half-to-even (torch.round / Python round)."""
    return np.rint(x)


def level_shapes(imshape, config):
    """Shapes of the reference-side (circularly padded) and moving-side pyramid levels and the
    tile grids, fine-to-coarse (reference alignment.py:27-37, utils_image.py:380-391)."""
    from .pyramid import gaussian_taps

    Ts = config.block_matching.tuning.tile_size
    factors = config.block_matching.tuning.factors
    tss = config.block_matching.tuning.tile_sizes
    h, w = imshape
    ph = h + (Ts - h % Ts) * (h % Ts != 0)
    pw = w + (Ts - w % Ts) * (w % Ts != 0)
    ref, mov = [], []
    rs, ms = (ph, pw), (h, w)
    for f in factors:
        if f != 1:
            r = gaussian_taps(f)[1]
            rs = ((rs[0] - 2 * r) // f, (rs[1] - 2 * r) // f)
            ms = ((ms[0] - 2 * r) // f, (ms[1] - 2 * r) // f)
        ref.append(rs)
        mov.append(ms)
    tiles = [(s[0] // ts, s[1] // ts) for s, ts in zip(ref, tss)]
    return ref, mov, tiles


# ----------------------------------------------------------------------------- ICA precompute
def init_ica(lvl, ts):
    """Gradients by the un-normalised [-1,0,1] filter with zero border and the per-tile 2x2
    Hessian (reference ICA.py:15-76; D9: no 1/2 factor)."""
    lvl = np.asarray(lvl, dtype=F32)
    H, W = lvl.shape
    gx = np.zeros_like(lvl)
    gy = np.zeros_like(lvl)
    p = np.pad(lvl, 1)
    gx[:] = p[1:-1, 2:] - p[1:-1, :-2]
    gy[:] = p[2:, 1:-1] - p[:-2, 1:-1]
    ny, nx = H // ts, W // ts
    tx = gx[: ny * ts, : nx * ts].reshape(ny, ts, nx, ts).transpose(0, 2, 1, 3).reshape(ny, nx, -1)
    ty = gy[: ny * ts, : nx * ts].reshape(ny, ts, nx, ts).transpose(0, 2, 1, 3).reshape(ny, nx, -1)
    hess = np.zeros((ny, nx, 2, 2), dtype=F32)
    # sequential float32 accumulation over the tile, row-major (ICA.py:57-70)
    a = np.zeros((ny, nx), F32)
    b = np.zeros((ny, nx), F32)
    c = np.zeros((ny, nx), F32)
    for k in range(ts * ts):
        a += tx[..., k] * tx[..., k]
        b += tx[..., k] * ty[..., k]
        c += ty[..., k] * ty[..., k]
    hess[..., 0, 0] = a
    hess[..., 0, 1] = b
    hess[..., 1, 0] = b
    hess[..., 1, 1] = c
    return gx, gy, hess


def init_alignment(ref_grey, config):
    """reference alignment.py:20-72.  Returns (pyramid, gradx, grady, hessian) lists COARSE FIRST.
    The FFT of the zero-padded reference tiles (alignment.py:56-61) is an implementation detail of the
    reference's correlation and has no counterpart here."""
    ref_grey = np.asarray(ref_grey, dtype=F32)
    h, w = ref_grey.shape
    Ts = config.block_matching.tuning.tile_size
    tss = config.block_matching.tuning.tile_sizes
    pb = (Ts - h % Ts) * (h % Ts != 0)
    pr = (Ts - w % Ts) * (w % Ts != 0)
    padded = np.pad(ref_grey, ((0, pb), (0, pr)), mode="wrap")  # F.pad 'circular', bottom/right
    factors = config.block_matching.tuning.factors
    pyr = build_gaussian_pyramid(padded, factors)
    gxs, gys, hs = [], [], []
    for i, lvl in enumerate(pyr):
        ts = tss[len(factors) - i - 1]
        gx, gy, hh = init_ica(lvl, ts)
        gxs.append(gx)
        gys.append(gy)
        hs.append(hh)
    return pyr, gxs, gys, hs


# ----------------------------------------------------------------------------- block matching
def _windows(mov, flow_int, ts, r, clamp):
    """Gather the (ts+2r)^2 search window of every tile.  Origin = tile*ts + flow_int - r.
    clamp=True: clamp-to-edge (block_matching.py:369-371); False: zero outside (:131-139)."""
    ny, nx = flow_int.shape[:2]
    h, w = mov.shape
    P = ts + 2 * r
    top = np.arange(ny)[:, None] * ts + flow_int[..., 1] - r
    left = np.arange(nx)[None, :] * ts + flow_int[..., 0] - r
    off = np.arange(P)
    yy = top[:, :, None, None] + off[None, None, :, None]
    xx = left[:, :, None, None] + off[None, None, None, :]
    yy, xx = np.broadcast_arrays(yy, xx)
    if clamp:
        return mov[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
    inb = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
    win = mov[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
    return np.where(inb, win, F32(0))


def _ref_tiles(ref, ny, nx, ts):
    return ref[: ny * ts, : nx * ts].reshape(ny, ts, nx, ts).transpose(0, 2, 1, 3)


def bm_l2(ref_lvl, mov_lvl, flow, ts, r, return_cost=False):
    """reference block_matching.py:20-76, 348-377.

    argmin over (2r+1)^2 shifts of  sum(win^2) - 2 sum(ref*win)  (= SSD - |ref|^2), first minimum in
    row-major (dy-major) order; the integer shift is ADDED to the un-rounded flow (:75-76).  The
    reference evaluates the correlation by FFT in float32; this restatement sums directly in
    float64, so it returns the exact argmin (near-ties may differ from the FFT path)."""
    ref_lvl = np.asarray(ref_lvl, dtype=F32)
    mov_lvl = np.asarray(mov_lvl, dtype=F32)
    flow = np.array(flow, dtype=F32)
    ny, nx = flow.shape[:2]
    fi = _rne(flow).astype(np.int64)
    win = _windows(mov_lvl, fi, ts, r, clamp=True).astype(np.float64)
    reft = _ref_tiles(ref_lvl, ny, nx, ts).astype(np.float64)
    n = 2 * r + 1
    cost = np.empty((ny, nx, n, n))
    for dy in range(n):
        for dx in range(n):
            wv = win[:, :, dy : dy + ts, dx : dx + ts]
            cost[:, :, dy, dx] = (wv * wv).sum((-1, -2)) - 2 * (reft * wv).sum((-1, -2))
    idx = cost.reshape(ny, nx, -1).argmin(-1)
    flow[..., 0] += (idx % n - r).astype(F32)
    flow[..., 1] += (idx // n - r).astype(F32)
    if return_cost:
        return flow, cost
    return flow


def bm_l1(ref_lvl, mov_lvl, flow, ts, r, effective=False, return_cost=False):
    """INTENDED semantics of reference block_matching.py:78-345 (the upstream kernels are
    undefined behaviour, SURVEY.md App. A D1 — this function is the build's specification, not a
    captured reference result): SAD over the tile, window origin tile*ts + round(flow) - r,
    zero outside the moving image, first minimum row-major, flow <- round(flow) + shift (D17).
    ``effective=True``: the most likely on-hardware outcome, flow <- round_half_even(flow)."""
    ref_lvl = np.asarray(ref_lvl, dtype=F32)
    mov_lvl = np.asarray(mov_lvl, dtype=F32)
    flow = np.array(flow, dtype=F32)
    ny, nx = flow.shape[:2]
    fr = _rne(flow)
    if effective:
        return fr.astype(F32)
    fi = fr.astype(np.int64)
    win = _windows(mov_lvl, fi, ts, r, clamp=False).astype(np.float64)
    reft = _ref_tiles(ref_lvl, ny, nx, ts).astype(np.float64)
    n = 2 * r + 1
    cost = np.empty((ny, nx, n, n))
    for dy in range(n):
        for dx in range(n):
            cost[:, :, dy, dx] = np.abs(reft - win[:, :, dy : dy + ts, dx : dx + ts]).sum((-1, -2))
    idx = cost.reshape(ny, nx, -1).argmin(-1)
    out = np.empty_like(flow)
    out[..., 0] = fr[..., 0] + (idx % n - r)
    out[..., 1] = fr[..., 1] + (idx // n - r)
    if return_cost:
        return out, cost
    return out


# ----------------------------------------------------------------------------- ICA
def ica(ref_lvl, gx, gy, hess, mov_lvl, flow, ts, n_iter, ica64_row_bug=True):
    """Lucas-Kanade inverse-compositional iterations per tile, float32 (reference ICA.py:78-482).

    Sampling: integer part by truncation toward zero, signed fraction from modf (D11); ts=8 clamps
    coordinates to the moving image (ICA.py:152-156), ts>=16 reads zero outside (:240-243); ts=64
    reproduces the row off-by-one of ica_kernel_64 (:437-449, D2) unless ``ica64_row_bug`` is False.
    Tiles with |det H| < 1e-10 are left untouched."""
    ref_lvl = np.asarray(ref_lvl, dtype=F32)
    mov = np.asarray(mov_lvl, dtype=F32)
    flow = np.array(flow, dtype=F32)
    ny, nx = flow.shape[:2]
    h, w = mov.shape
    hess = np.asarray(hess, dtype=F32)
    A00, A01, A10, A11 = hess[..., 0, 0], hess[..., 0, 1], hess[..., 1, 0], hess[..., 1, 1]
    det = A00 * A11 - A01 * A10
    ok = ~(np.abs(det) < F32(1e-10))
    with np.errstate(all="ignore"):
        det_inv = F32(1.0) / det
    reft = _ref_tiles(ref_lvl, ny, nx, ts)
    gxt = _ref_tiles(np.asarray(gx, F32), ny, nx, ts)
    gyt = _ref_tiles(np.asarray(gy, F32), ny, nx, ts)
    yy = (np.arange(ny)[:, None, None, None] * ts + np.arange(ts)[None, None, :, None])
    xx = (np.arange(nx)[None, :, None, None] * ts + np.arange(ts)[None, None, None, :])
    sx = flow[..., 0].copy()
    sy = flow[..., 1].copy()

    def fetch(y, x):
        if ts == 8:
            return mov[y, x]  # already clamped
        inb = (y >= 0) & (y < h) & (x >= 0) & (x < w)
        return np.where(inb, mov[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1)], F32(0))

    for _ in range(n_iter):
        ix = np.trunc(sx).astype(np.int64)[:, :, None, None]
        iy = np.trunc(sy).astype(np.int64)[:, :, None, None]
        fx = (sx - np.trunc(sx)).astype(F32)[:, :, None, None]
        fy = (sy - np.trunc(sy)).astype(F32)[:, :, None, None]
        x0 = xx + ix
        y0 = yy + iy
        x0, y0 = np.broadcast_arrays(x0, y0)
        if ts == 8:
            x0 = np.clip(x0, 0, w - 1)
            y0 = np.clip(y0, 0, h - 1)
            x1 = np.clip(x0 + 1, 0, w - 1)
            y1 = np.clip(y0 + 1, 0, h - 1)
            ytop, ybot = y0, y1
        else:
            x1 = x0 + 1
            ytop, ybot = y0, y0 + 1
            if ts == 64 and ica64_row_bug:
                # rows (y0, y0+2) for the first of each thread's 4 rows, (y+1, y+2) for the others
                first = (np.arange(ts) % 4 == 0)[None, None, :, None]
                ytop = np.where(first, y0, y0 + 1)
                ybot = y0 + 2
        m00 = fetch(ytop, x0)
        m01 = fetch(ytop, x1)
        m10 = fetch(ybot, x0)
        m11 = fetch(ybot, x1)
        top = m00 + (m01 - m00) * fx
        bot = m10 + (m11 - m10) * fx
        interp = top + (bot - top) * fy
        gradt = interp - reft
        B0 = (-gxt * gradt).astype(F32).sum((-1, -2), dtype=F32)
        B1 = (-gyt * gradt).astype(F32).sum((-1, -2), dtype=F32)
        with np.errstate(all="ignore"):
            ux = det_inv * (A11 * B0 - A01 * B1)
            uy = det_inv * (-A10 * B0 + A00 * B1)
        sx = np.where(ok, sx + ux, sx).astype(F32)
        sy = np.where(ok, sy + uy, sy).astype(F32)
    flow[..., 0] = sx
    flow[..., 1] = sy
    return flow


# ----------------------------------------------------------------------------- flow upscaling
def _cubic_w(t, A=-0.75):
    t = np.asarray(t, dtype=np.float64)

    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], -1)


def _interp_axis(a, k, mode, axis):
    """torch F.interpolate(scale_factor=k, align_corners=False) along one axis (float32 weights)."""
    n = a.shape[axis]
    dst = np.arange(n * k)
    if mode == "nearest":
        return np.take(a, np.floor(dst * (1.0 / k)).astype(np.int64).clip(0, n - 1), axis=axis)
    src = (dst + 0.5) / k - 0.5
    shape = [1] * a.ndim
    shape[axis] = -1
    if mode == "bilinear":
        src = np.maximum(src, 0.0)
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n - 1)
        l1 = (src - i0).astype(F32).reshape(shape)
        l0 = (F32(1) - l1)
        return (np.take(a, i0, axis=axis) * l0 + np.take(a, i1, axis=axis) * l1).astype(F32)
    if mode == "bicubic":
        i0 = np.floor(src).astype(np.int64)
        t = src - i0
        wts = _cubic_w(t).astype(F32)
        out = np.zeros(np.take(a, i0.clip(0, n - 1), axis=axis).shape, dtype=F32)
        for j in range(4):
            idx = (i0 - 1 + j).clip(0, n - 1)
            out = out + np.take(a, idx, axis=axis) * wts[:, j].reshape(shape)
        return out.astype(F32)
    raise ValueError(mode)


def upscale_lvl(flow, npatchs, l, config):
    """reference alignment.py:150-172: repeat/interpolate by factors[l+1] // (ts[l] // ts[l+1]),
    multiply by factors[l+1], zero-pad bottom/right to the new tile grid."""
    bm = config.block_matching.tuning
    new_ts, prev_ts = bm.tile_sizes[l], bm.tile_sizes[l + 1]
    up = bm.factors[l + 1]
    k = up // (new_ts // prev_ts)
    mode = bm.flow_upscale_mode
    f = np.asarray(flow, dtype=F32)
    if mode == "bicubic":
        # torch's 2-D bicubic applies x-weights inside y-weights: sum_y wy * (sum_x wx * v)
        f = _interp_axis(f, k, mode, 1)
        f = _interp_axis(f, k, mode, 0)
    else:
        f = _interp_axis(f, k, mode, 0)
        f = _interp_axis(f, k, mode, 1)
    f = (f * F32(up)).astype(F32)
    py, px = npatchs[0] - f.shape[0], npatchs[1] - f.shape[1]
    if py > 0 or px > 0:
        f = np.pad(f, ((0, max(py, 0)), (0, max(px, 0)), (0, 0)))
    return f


# ----------------------------------------------------------------------------- driver
def align_lvl(ref_lvl, gx, gy, hess, mov_lvl, flow, l, config):
    """reference alignment.py:125-147."""
    bm = config.block_matching.tuning
    ts, r, metric = bm.tile_sizes[l], bm.search_radii[l], bm.metrics[l]
    if metric == "L2":
        flow = bm_l2(ref_lvl, mov_lvl, flow, ts, r)
    elif metric == "L1":
        flow = bm_l1(ref_lvl, mov_lvl, flow, ts, r)
    elif metric == "L1_ref_effective":
        flow = bm_l1(ref_lvl, mov_lvl, flow, ts, r, effective=True)
    else:
        raise ValueError("Unknown block matching metric {}".format(metric))
    bug = bool(config.get("compat", {}).get("ica64_row_bug", True)) if hasattr(config, "get") else True
    return ica(ref_lvl, gx, gy, hess, mov_lvl, flow, ts, config.ica.tuning.n_iter, ica64_row_bug=bug)


def align(ref_pyr, ref_gx, ref_gy, ref_hess, img_grey, config):
    """reference alignment.py:84-123: coarse-to-fine over the pyramid.  Note the moving image is
    NOT padded (D16) so its levels can be smaller than the reference's."""
    factors = config.block_matching.tuning.factors
    mov_pyr = build_gaussian_pyramid(np.asarray(img_grey, dtype=F32), factors)
    flow = None
    n = len(ref_pyr)
    for i in range(n):
        l = n - i - 1
        ts = config.block_matching.tuning.tile_sizes[l]
        grid = (ref_pyr[i].shape[0] // ts, ref_pyr[i].shape[1] // ts)
        if flow is None:
            flow = np.zeros((*grid, 2), dtype=F32)
        else:
            flow = upscale_lvl(flow, grid, l, config)
        flow = align_lvl(ref_pyr[i], ref_gx[i], ref_gy[i], ref_hess[i], mov_pyr[i], flow, l, config)
    return flow

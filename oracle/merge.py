"""Oracle: Alg. 4 / Alg. 11 accumulation and normalisation (restates reference merge.py:22-434,
linalg.py:38-200, utils.py:62-120, utils_image.py:311-325).

Test infrastructure.  Numba typing is followed: coordinates and weights are
float64, the per-pixel ``val``/``acc`` accumulators are float32 and are rounded after every tap."""
import numpy as np

F32 = np.float32
F64 = np.float64


def _pymax0(z):
    """max(0, z) with Python/Numba semantics: NaN -> 0."""
    return np.where(z > 0, z, 0.0)


def merge(comp, flow, covs, r, num, den, cfa, config):
    """Alg. 4 (merge.py:236-434).  ``num``/``den`` float32[sH, sW, 3] are updated in place.  Grey mode: every tap goes
    to channel 0 (channels 1, 2 stay untouched) and the covariance grid is the pixel grid (merge.py:349-354, 410)."""
    bayer = config.mode == "bayer"
    comp = np.asarray(comp, dtype=F32)
    flow = np.asarray(flow, dtype=F32)
    covs = np.asarray(covs, dtype=F32)
    r = np.asarray(r, dtype=F32)
    cfa = np.asarray(cfa).astype(np.int64)
    scale = F64(config.scale)
    iso = config.merging.kernel == "iso"
    ts = config.block_matching.tuning.tile_size
    hr_h, hr_w, _ = num.shape
    lr_h, lr_w = comp.shape
    hi = np.arange(hr_h)[:, None] + 0 * np.arange(hr_w)[None, :]
    hj = 0 * np.arange(hr_h)[:, None] + np.arange(hr_w)[None, :]
    lr_x = (hj + 0.5) / scale
    lr_y = (hi + 0.5) / scale
    px = (lr_x // ts).astype(np.int64)
    py = (lr_y // ts).astype(np.int64)
    flowx = flow[py, px, 0].astype(F64)
    flowy = flow[py, px, 1].astype(F64)
    i_r = np.minimum(lr_y.astype(np.int64), lr_h - 1)
    j_r = np.minimum(lr_x.astype(np.int64), lr_w - 1)
    local_r = r[i_r, j_r].astype(F64)
    mx = lr_x + flowx
    my = lr_y + flowy
    inb = (mx >= 0) & (mx < lr_w) & (my >= 0) & (my < lr_h)
    mxs = np.where(inb, mx, 0.0)
    mys = np.where(inb, my, 0.0)
    with np.errstate(all="ignore"):
        if not iso:
            kj = mxs / 2 - 0.5 if bayer else mxs - 0.5
            ki = mys / 2 - 0.5 if bayer else mys - 0.5
            fx = kj - np.trunc(kj)
            fy = ki - np.trunc(ki)
            x0 = np.maximum(np.trunc(kj).astype(np.int64), 0)
            y0 = np.maximum(np.trunc(ki).astype(np.int64), 0)
            x1 = np.minimum(x0 + 1, covs.shape[1] - 1)
            y1 = np.minimum(y0 + 1, covs.shape[0] - 1)

            def interp(a, b):
                tr, tl = covs[y0, x0, a, b], covs[y0, x1, a, b]
                br, bl = covs[y1, x0, a, b], covs[y1, x1, a, b]
                top = tr.astype(F64) + fx * (tl - tr).astype(F64)  # float32 difference, float64 lerp
                bot = br.astype(F64) + fx * (bl - br).astype(F64)
                return top + fy * (bot - top)

            cxx, cxy, cyy = interp(0, 0), interp(0, 1), interp(1, 1)
            det = cxx * cyy - cxy * cxy
            inv_det = 1.0 / det
            ixx, ixy, iyy = inv_det * cyy, -inv_det * cxy, inv_det * cxx
        cj = np.trunc(mxs).astype(np.int64)
        ci = np.trunc(mys).astype(np.int64)
        mj = mxs - 0.5
        mi = mys - 0.5
        val = np.zeros((3, hr_h, hr_w), F32)
        acc = np.zeros((3, hr_h, hr_w), F32)
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                j = cj + dj
                i = ci + di
                ok = inb & (j >= 0) & (j < lr_w) & (i >= 0) & (i < lr_h)
                jc = np.clip(j, 0, lr_w - 1)
                ic = np.clip(i, 0, lr_h - 1)
                ch = cfa[ic % 2, jc % 2] if bayer else np.zeros_like(ic)
                c = comp[ic, jc].astype(F64)
                dx = j - mj
                dy = i - mi
                if iso:
                    z = 2 * (dx * dx + dy * dy)
                else:
                    z = ixx * dx * dx + 2 * ixy * dx * dy + iyy * dy * dy
                z = _pymax0(z)
                w = np.exp(-0.5 * z)
                wr = w * local_r
                for k in range(3):
                    m = ok & (ch == k)
                    val[k] = np.where(m, (val[k].astype(F64) + wr * c).astype(F32), val[k])
                    acc[k] = np.where(m, (acc[k].astype(F64) + wr).astype(F32), acc[k])
    for k in range(3 if bayer else 1):
        num[..., k] = np.where(inb, num[..., k] + val[k], num[..., k])
        den[..., k] = np.where(inb, den[..., k] + acc[k], den[..., k])


def merge_ref(ref, covs, num, den, cfa, config, acc_rob=None):
    """Alg. 11 (merge.py:22-233).  Position = idx/scale without the half-pixel offset
    (D7), stored in float32; centre tap = round-half-even; covariance via interpolate_cov +
    invert_2x2 (identity when |det| <= 1e-10 or NaN).  Grey mode: channel 0 only, covariance read at the position
    itself (merge.py:131-137, 191-194)."""
    bayer = config.mode == "bayer"
    ref = np.asarray(ref, dtype=F32)
    covs = np.asarray(covs, dtype=F32)
    cfa = np.asarray(cfa).astype(np.int64)
    scale = F64(config.scale)
    iso = config.merging.kernel == "iso"
    denoise = bool(config.accumulated_robustness_denoiser.enabled)
    oh, ow, _ = num.shape
    H, W = ref.shape
    oi = np.arange(oh)[:, None] + 0 * np.arange(ow)[None, :]
    oj = 0 * np.arange(oh)[:, None] + np.arange(ow)[None, :]
    py = (oi / scale).astype(F32)  # coarse_ref_sub_pos is a float32 local array
    px = (oj / scale).astype(F32)
    with np.errstate(all="ignore"):
        if not iso:
            gy = ((py.astype(F64) - 0.5) / 2).astype(F32) if bayer else py
            gx = ((px.astype(F64) - 0.5) / 2).astype(F32) if bayer else px
            x0 = np.maximum(np.floor(gx), 0).astype(np.int64)
            y0 = np.maximum(np.floor(gy), 0).astype(np.int64)
            x1 = np.minimum(x0 + 1, covs.shape[1] - 1)
            y1 = np.minimum(y0 + 1, covs.shape[0] - 1)
            rx = (gx - np.trunc(gx)).astype(F64)  # modf fraction (signed)
            ry = (gy - np.trunc(gy)).astype(F64)

            def interp(a, b):
                c00, c01 = covs[y0, x0, a, b].astype(F64), covs[y0, x1, a, b].astype(F64)
                c10, c11 = covs[y1, x0, a, b].astype(F64), covs[y1, x1, a, b].astype(F64)
                return (c00 * (1 - rx) * (1 - ry) + c01 * rx * (1 - ry) + c10 * (1 - rx) * ry
                        + c11 * rx * ry).astype(F32)

            m00, m01, m10, m11 = interp(0, 0), interp(0, 1), interp(1, 0), interp(1, 1)
            det = m00 * m11 - m01 * m10  # float32
            good = np.abs(det) > 1e-10
            det_i = 1 / det.astype(F64)
            i00 = np.where(good, (m11.astype(F64) * det_i).astype(F32), F32(1))
            i01 = np.where(good, (-m01.astype(F64) * det_i).astype(F32), F32(0))
            i10 = np.where(good, (-m10.astype(F64) * det_i).astype(F32), F32(0))
            i11 = np.where(good, (m00.astype(F64) * det_i).astype(F32), F32(1))
        if denoise:
            d = config.accumulated_robustness_denoiser.merge
            ry_i = np.minimum(np.rint(py).astype(np.int64), acc_rob.shape[0] - 1)
            rx_i = np.minimum(np.rint(px).astype(np.int64), acc_rob.shape[1] - 1)
            lacc = np.asarray(acc_rob)[ry_i, rx_i]
            low = lacc <= d.max_frame_count
            power = np.where(low, F64(d.max_multiplier), 1.0)
            rad_map = np.where(low, int(d.rad_max), 1)
            rmax = max(int(d.rad_max), 1)
        else:
            power = 1.0
            rad_map = np.ones((oh, ow), np.int64)
            rmax = 1
        cx = np.rint(px).astype(np.int64)
        cy = np.rint(py).astype(np.int64)
        val = np.zeros((3, oh, ow), F32)
        acc = np.zeros((3, oh, ow), F32)
        for i in range(-rmax, rmax + 1):
            for j in range(-rmax, rmax + 1):
                pj = cx + j
                pi = cy + i
                ok = (np.abs(i) <= rad_map) & (np.abs(j) <= rad_map) & (pj >= 0) & (pj < W) & (pi >= 0) & (pi < H)
                jc = np.clip(pj, 0, W - 1)
                ic = np.clip(pi, 0, H - 1)
                ch = cfa[ic % 2, jc % 2] if bayer else np.zeros_like(ic)
                c = ref[ic, jc].astype(F64)
                dx = pj - px.astype(F64)
                dy = pi - py.astype(F64)
                if iso:
                    y = _pymax0(2 * (dx * dx + dy * dy))
                else:
                    q = (i00.astype(F64) * dx * dx + dx * dy * (i01 + i10).astype(F64) + i11.astype(F64) * dy * dy)
                    y = _pymax0(q)
                y = y / power
                w = np.exp(-0.5 * y)
                for k in range(3):
                    m = ok & (ch == k)
                    val[k] = np.where(m, (val[k].astype(F64) + c * w).astype(F32), val[k])
                    acc[k] = np.where(m, (acc[k].astype(F64) + w).astype(F32), acc[k])
    if denoise:
        over = lacc < d.max_frame_count
    else:
        over = np.zeros((oh, ow), bool)
    for k in range(3 if bayer else 1):
        num[..., k] = np.where(over, val[k], num[..., k] + val[k])
        den[..., k] = np.where(over, acc[k], den[..., k] + acc[k])


def divide(num, den):
    """utils.py:62-90: num /= den in float32 (0/0 stays NaN)."""
    with np.errstate(all="ignore"):
        num[...] = num / den
    return num

"""Oracle: Alg. 6-9 robustness (restates reference robustness.py:23-690, utils_image.py:395-406).

Test infrastructure.  Numba typing is followed; "stds" are variances."""
import numpy as np

F32 = np.float32
F64 = np.float64


def guide_image(raw, cfa, wb):
    """Alg. 7 (robustness.py:207-225): float32[3, H/2, W/2]; divisions by wb in float64."""
    raw = np.asarray(raw, dtype=F32)
    cfa = np.asarray(cfa).astype(np.int64)
    wb = np.asarray(wb, dtype=F64)
    h, w = raw.shape[0] // 2, raw.shape[1] // 2
    guide = np.zeros((3, h, w), F32)
    g = np.zeros((h, w), F64)
    for i in range(2):
        for j in range(2):
            c = int(cfa[i, j])
            x = raw[i : 2 * h : 2, j : 2 * w : 2].astype(F64) / wb[c]
            if c == 1:
                g = g + x
            else:
                guide[c] = x.astype(F32)
    guide[1] = (g / 2).astype(F32)
    return guide


def local_stats(guide):
    """Alg. 8 (robustness.py:269-294): 3x3 clamp-border mean and variance; float32 running sums in
    (i, j) order, the two divisions by 9 in float64."""
    guide = np.asarray(guide, dtype=F32)
    nc, h, w = guide.shape
    p = np.pad(guide, ((0, 0), (1, 1), (1, 1)), mode="edge")
    s0 = np.zeros_like(guide)
    s1 = np.zeros_like(guide)
    for i in range(3):
        for j in range(3):
            v = p[:, i : i + h, j : j + w]
            s0 += v
            s1 += v * v
    mean = s0.astype(F64) / 9
    var = s1.astype(F64) / 9 - mean * mean
    return mean.astype(F32), var.astype(F32)


def _dodgson(x):
    """utils_image.py:399-406 on float64."""
    a = np.abs(x)
    return np.where(a <= 0.5, -2 * a * a + 1, np.where(a <= 1.5, a * a - 5 / 2 * a + 1.5, 0.0))


def upscale_warp_stats(stats, tile_size=None, flow=None):
    """robustness.py:296-421: guide-res map -> raw res (s = 2 hard-coded, D5), optionally warped by the
    per-tile flow; 3x3 Dodgson taps around round-half-even(LR) with clamped indices, normalised;
    +inf where the guide position is outside [0, n) (D6: row 0 / column 0 of the un-warped ref map).

    A 1-channel map (grey mode) keeps its size (robustness.py:337-343) while the kernel still divides the position by
    its hard-coded s = 2 (robustness.py:358): the output is the top-left quadrant of the map stretched over the whole
    frame, for the reference frame and the warped frames alike.  Deterministic, so reproduced (D5)."""
    stats = np.asarray(stats, dtype=F32)
    nc, lh, lw = stats.shape
    H, W = (2 * lh, 2 * lw) if nc == 3 else (lh, lw)
    y = np.arange(H)[:, None]
    x = np.arange(W)[None, :]
    if flow is None:
        fx = np.zeros((H, W), F64)
        fy = np.zeros((H, W), F64)
    else:
        flow = np.asarray(flow, dtype=F32)
        ty = (y // tile_size) + 0 * x
        tx = (x // tile_size) + 0 * y
        fx = flow[ty, tx, 0].astype(F64)
        fy = flow[ty, tx, 1].astype(F64)
    ly = (y + fy + 0.5) / 2 - 0.5
    lx = (x + fx + 0.5) / 2 - 0.5
    oob = ~((ly >= 0) & (ly < lh) & (lx >= 0) & (lx < lw))
    lyc = np.where(oob, 0.0, ly)
    lxc = np.where(oob, 0.0, lx)
    cy = np.rint(lyc).astype(np.int64)
    cx = np.rint(lxc).astype(np.int64)
    buf = np.zeros((nc, H, W), F32)
    wacc = np.zeros((H, W), F64)
    for i in (-1, 0, 1):
        y_ = np.clip(cy + i, 0, lh - 1)
        wy = _dodgson(y_ - lyc)
        for j in (-1, 0, 1):
            x_ = np.clip(cx + j, 0, lw - 1)
            wgt = wy * _dodgson(x_ - lxc)
            buf = (buf.astype(F64) + stats[:, y_, x_].astype(F64) * wgt).astype(F32)
            wacc = wacc + wgt
    with np.errstate(all="ignore"):
        out = (buf.astype(F64) / wacc).astype(F32)
    out[:, oob] = np.inf
    return out


def apply_noise_model(d_p, ref_means, ref_vars, std_curve, diff_curve):
    """robustness.py:505-528.  The curve index of a non-finite brightness (D6 border) is taken as 0 —
    the reference reads out of bounds there and the value is discarded by the NaN -> 0 clamp."""
    std_curve = np.asarray(std_curve, F64)
    diff_curve = np.asarray(diff_curve, F64)
    nc = ref_means.shape[0]
    d_sq = np.zeros(ref_means.shape[1:], F64)
    s_sq = np.zeros(ref_means.shape[1:], F64)
    with np.errstate(all="ignore"):
        for c in range(nc):
            b = ref_means[c].astype(F64)
            idx = np.where(np.isfinite(b), np.rint(1000 * b), 0).astype(np.int64)
            idx = np.clip(idx, 0, len(std_curve) - 1)
            d_t = diff_curve[idx]
            s_t = std_curve[idx]
            sp = ref_vars[c].astype(F64)
            st2 = s_t * s_t
            s_sq = s_sq + np.where(st2 > sp, st2, sp)  # max(sigma_p_sq, sigma_t^2): b if b > a else a
            dp = d_p[c]
            dp2 = (dp * dp).astype(F32).astype(F64)
            shrink = dp2 / (dp2 + d_t * d_t)
            d_sq = d_sq + dp2 * shrink * shrink
    return d_sq.astype(F32), s_sq.astype(F32)


def compute_s(flow, M_th, s1, s2):
    """robustness.py:570-612: 3x3 tile-neighbourhood flow spread."""
    flow = np.asarray(flow, dtype=F32)
    ny, nx = flow.shape[:2]
    mx = np.full((ny, nx, 2), -np.inf, F32)
    mn = np.full((ny, nx, 2), np.inf, F32)
    for i in (-1, 0, 1):
        for j in (-1, 0, 1):
            ys, ye = max(0, -i), min(ny, ny - i)
            xs, xe = max(0, -j), min(nx, nx - j)
            src = flow[ys + i : ye + i, xs + j : xe + j]
            mx[ys:ye, xs:xe] = np.maximum(mx[ys:ye, xs:xe], src)
            mn[ys:ye, xs:xe] = np.minimum(mn[ys:ye, xs:xe], src)
    d = mx - mn
    m = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(F32)
    return np.where(m.astype(F64) > F64(M_th) * F64(M_th), F32(s1), F32(s2)).astype(F32)


def robustness_threshold(d_sq, s_sq, S, t, tile_size):
    """robustness.py:627-639: R = clamp(S[tile] * exp(-d^2/sigma^2) - t, 0, 1); NaN -> 0."""
    H, W = d_sq.shape
    ty = np.arange(H)[:, None] // tile_size
    tx = np.arange(W)[None, :] // tile_size
    with np.errstate(all="ignore"):
        e = np.exp((-d_sq / s_sq).astype(F32)).astype(F32)
        v = (S[ty, tx] * e).astype(F32).astype(F64) - F64(t)
    v = np.where(v > 0, v, 0.0)
    v = np.where(v < 1, v, 1.0)
    return v.astype(F32)


def local_min(R):
    """Alg. 9 (robustness.py:670-686): 5x5 clamp-border minimum."""
    R = np.asarray(R, dtype=F32)
    H, W = R.shape
    p = np.pad(R, 2, mode="edge")
    out = np.full((H, W), np.inf, F32)
    for i in range(5):
        for j in range(5):
            out = np.minimum(out, p[i : i + H, j : j + W])
    return out


def _guide(raw, cfa, wb, config):
    """Bayer: Alg. 7; grey: the frame itself as one channel, white balance not involved (robustness.py:62-66, 145-148)."""
    if config.mode == "bayer":
        return guide_image(raw, cfa, wb)
    return np.asarray(raw, dtype=F32)[None]


def init_robustness(ref, cfa, wb, config):
    """robustness.py:23-76: reference-frame local means / variances at raw resolution."""
    if not config.robustness.enabled:
        return None, None
    m, v = local_stats(_guide(ref, cfa, wb, config))
    return upscale_warp_stats(m), upscale_warp_stats(v)


def compute_robustness(comp, ref_means, ref_vars, flow, cfa, wb, noise_model, config, debug=None, S=None):
    """Alg. 6 (robustness.py:79-170).  `S`: precomputed flow-irregularity weights for `flow` (the multi-GPU tests pass
    the rows of the full frame's map that belong to a sub-image: S is the one stage that is not row-local)."""
    comp = np.asarray(comp, dtype=F32)
    if not config.robustness.enabled:
        return np.ones_like(comp, F32)
    ts = config.block_matching.tuning.tile_size
    t = config.robustness.tuning
    cm, _ = local_stats(_guide(comp, cfa, wb, config))
    cmu = upscale_warp_stats(cm, ts, flow)
    with np.errstate(all="ignore"):
        d_p = np.abs(ref_means - cmu).astype(F32)
    d_sq, s_sq = apply_noise_model(d_p, ref_means, ref_vars, noise_model[0], noise_model[1])
    if S is None:
        S = compute_s(flow, t.Mt, t.s1, t.s2)
    R = robustness_threshold(d_sq, s_sq, S, t.t, ts)
    if debug is not None:
        debug.update(comp_means_up=cmu, d_sq=d_sq, sigma_sq=s_sq, S=S, R=R)
    return local_min(R)

"""Oracle: Alg. 5 kernel covariances (restates reference kernels.py:29-243, linalg.py:87-185,
utils_image.py:117-170).  Test infrastructure.  Numba typing is followed."""
import numpy as np

from .grey import decimate_to_grey

F32 = np.float32
F64 = np.float64


def gat(img, alpha, beta):
    """Generalised Anscombe transform, float64 maths stored float32 (utils_image.py:157-170)."""
    v = F64(alpha) * np.asarray(img, dtype=F32).astype(F64) + 3 / 8 * F64(alpha) * F64(alpha) + F64(beta)
    v = np.where(v > 0, v, 0.0)  # max(0, VST)
    return (2 / F64(alpha) * np.sqrt(v)).astype(F32)


def _gradients(grey):
    """Two chained float32 convs (kernels.py:97-116): gx = 1/2[-1,1] (x) 1/2[1,1]^T, gy the transpose."""
    g = grey
    t0 = F32(-0.5) * g[:, :-1] + F32(0.5) * g[:, 1:]
    t1 = F32(0.5) * g[:, :-1] + F32(0.5) * g[:, 1:]
    gx = F32(0.5) * t0[:-1, :] + F32(0.5) * t0[1:, :]
    gy = F32(-0.5) * t1[:-1, :] + F32(0.5) * t1[1:, :]
    return gx.astype(F32), gy.astype(F32)


def eigen_2x2(M00, M01, M10, M11):
    """linalg.py:87-185 on float32 arrays: eigenvalues (largest modulus first) and unit eigenvectors."""
    b = -(M00 + M11)  # f32
    c = M00 * M11 - M01 * M10  # f32
    bb = (b * b).astype(F32).astype(F64)
    delta = bb - 4 * c.astype(F64)
    delta = np.where(delta > 0, delta, 0.0)  # max(delta, 0); NaN -> 0 like Python max
    sq = np.sqrt(delta)
    r1 = (-b.astype(F64) + sq) / 2
    r2 = (-b.astype(F64) - sq) / 2
    swap = ~(np.abs(r1) >= np.abs(r2))
    l1 = np.where(swap, r2, r1).astype(F32)
    l2 = np.where(swap, r1, r2).astype(F32)
    # eigenvectors (linalg.py:152-179)
    ident = (M01 == 0) & (M00 == M11)
    a0 = (M00 + M01 - l2).astype(F32)
    a1 = (M10 + M11 - l2).astype(F32)
    with np.errstate(all="ignore"):
        nrm = np.sqrt((a0 * a0 + a1 * a1).astype(F32)).astype(F32)
        n0 = (a0 / nrm).astype(F32)
        n1 = (a1 / nrm).astype(F32)
    sign = np.copysign(1.0, n0)
    e1x, e1y = n0, n1
    e2y = np.abs(n0)
    e2x = (-n1 * sign).astype(F32)
    # elif e1[1] == 0: e1 = (1, 0), e2 = (0, 1)
    z1 = a1 == 0
    e1x = np.where(z1, F32(1), e1x)
    e1y = np.where(z1, a1, e1y)  # stays 0 (keeps the sign of zero, irrelevant downstream)
    e2x = np.where(z1, F32(0), e2x)
    e2y = np.where(z1, F32(1), e2y)
    # if e1[0] == 0: e1 = (0, 1), e2 = (1, 0)   (tested first in the reference)
    z0 = a0 == 0
    e1x = np.where(z0, a0, e1x)
    e1y = np.where(z0, F32(1), e1y)
    e2x = np.where(z0, F32(1), e2x)
    e2y = np.where(z0, F32(0), e2y)
    # multiple of identity
    e1x = np.where(ident, F32(1), e1x)
    e1y = np.where(ident, F32(0), e1y)
    e2x = np.where(ident, F32(0), e2x)
    e2y = np.where(ident, F32(1), e2y)
    return l1, l2, e1x.astype(F32), e1y.astype(F32), e2x.astype(F32), e2y.astype(F32)


def compute_k(l1, l2, t, law):
    """kernels.py:195-243.  Returns k1, k2 rounded to float32 (they are stored in a float32 local array)."""
    with np.errstate(all="ignore"):
        A = 1 + np.sqrt(((l1 - l2) / (l1 + l2)).astype(F32)).astype(F32).astype(F64)
        D = 1 - np.sqrt(l1).astype(F32).astype(F64) / F64(t.D_tr) + F64(t.D_th)
    D = np.where(D > 0, D, 0.0)  # max(0, x)
    D = np.where(D < 1, D, 1.0)  # min(1, .)
    if law == "hard_threshold":
        big = A > 1.95
        k1 = np.where(big, 1 / F64(t.k_shrink), 1.0)
        k2 = np.where(big, F64(t.k_stretch), 1.0)
    elif law == "linear":
        k1 = 1 + A / 2 * (1 / F64(t.k_shrink) - 1)
        k2 = 1 + A / 2 * (F64(t.k_stretch) - 1)
    else:
        raise ValueError(f"Unknown selection law: {law}")
    kd, kn = F64(t.k_detail), F64(t.k_denoise)
    return (kd * ((1 - D) * k1 + D * kn)).astype(F32), (kd * ((1 - D) * k2 + D * kn)).astype(F32)


def estimate_kernels(img, config):
    """reference kernels.py:29-137: returns covs float32[H/2, W/2, 2, 2] (bayer mode: one per quad, from the 2x2-mean
    decimation of the stabilised image) or float32[H, W, 2, 2] (grey mode: one per pixel, kernels.py:83-87)."""
    law = config.merging.selection_law
    if law not in ("hard_threshold", "linear"):
        raise ValueError(f"Unknown selection law: {law}")
    vst = gat(img, config.noise_model.alpha, config.noise_model.beta)
    grey = decimate_to_grey(vst) if config.mode == "bayer" else vst
    gh, gw = grey.shape
    gx, gy = _gradients(grey)
    # 2x2 window of gradient samples (y-1..y, x-1..x), zero where out of the [gh-1, gw-1] gradient grid;
    # float32 accumulation in the order (i, j) = (0,0), (0,1), (1,0), (1,1)  (kernels.py:159-175)
    pgx = np.zeros((gh + 1, gw + 1), F32)
    pgy = np.zeros((gh + 1, gw + 1), F32)
    pgx[1:gh, 1:gw] = gx
    pgy[1:gh, 1:gw] = gy
    T00 = np.zeros((gh, gw), F32)
    T01 = np.zeros((gh, gw), F32)
    T11 = np.zeros((gh, gw), F32)
    for i in range(2):
        for j in range(2):
            ax = pgx[i : i + gh, j : j + gw]
            ay = pgy[i : i + gh, j : j + gw]
            T00 += ax * ax
            T01 += ax * ay
            T11 += ay * ay
    l1, l2, e1x, e1y, e2x, e2y = eigen_2x2(T00, T01, T01, T11)
    k1, k2 = compute_k(l1, l2, config.merging.tuning, law)
    k1s = k1 * k1
    k2s = k2 * k2
    covs = np.empty((gh, gw, 2, 2), F32)
    with np.errstate(all="ignore"):
        covs[..., 0, 0] = k1s * e1x * e1x + k2s * e2x * e2x
        covs[..., 0, 1] = k1s * e1x * e1y + k2s * e2x * e2y
        covs[..., 1, 0] = covs[..., 0, 1]
        covs[..., 1, 1] = k1s * e1y * e1y + k2s * e2y * e2y
    return covs

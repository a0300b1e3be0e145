"""Deterministic synthetic RAW bursts.

The reference ships no sample burst (its ``test_burst/`` folder is empty) and
its DNG loader needs rawpy/exifread, so parity tests and ``bench.py`` use an
analytic scene whose sub-pixel translations are exact:

    I_c(y, x) = clip(0.5 + sum_k a_k sin(2 pi (u_k x + v_k y) + phi_{k,c}), 0.02, 0.98)

with K = 64 plane waves whose radial frequencies are log-uniform (1/f spectrum; SURVEY.md §8d's
uniform-frequency scene left the coarse pyramid levels textureless, so block matching had
nothing to lock onto — see ``scene_params``).

Frame ``n`` samples the scene at ``(x + dx_n, y + dy_n)``, is mosaicked with
the CFA, gets heteroscedastic Gaussian noise ``N(0, alpha I + beta)`` and is
clipped to [0, 1].  The sum of sinusoids is evaluated as a rank-2K matrix
product, so it is cheap with NumPy on the host and with torch on the GPU.

Only NumPy (and optionally torch) is imported here so that the module can be
loaded stand-alone by the test tooling.
"""
from __future__ import annotations

import numpy as np

# ISO-100 values quoted in the reference README (README.md:159-160)
ALPHA_ISO100 = 1.80710882e-4
BETA_ISO100 = 3.1937599182128e-6

N_WAVES = 64


def scene_params(seed=1234, n_waves=N_WAVES):
    """Plane waves with log-uniform radial frequency in [0.002, 0.25] cycles/pixel and uniform
    orientation: equal energy per octave (a natural-image-like 1/f amplitude spectrum), so every
    level of the alignment pyramid (down to 1/32 resolution) sees texture.  The three colour
    channels share the geometry and differ in phase jitter and gain (correlated channels, as in
    real scenes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rho = 0.002 * (0.25 / 0.002) ** rng.uniform(0.0, 1.0, n_waves)
    th = rng.uniform(0.0, 2.0 * np.pi, n_waves)
    u = rho * np.cos(th)
    v = rho * np.sin(th)
    a = np.full(n_waves, 0.16 * np.sqrt(2.0 / n_waves))
    phi0 = rng.uniform(0.0, 2.0 * np.pi, (n_waves, 1))
    phi = phi0 + rng.uniform(-0.4, 0.4, (n_waves, 3))
    return u, v, a, phi


def frame_shifts(n_frames, seed=1234, max_shift=4.0):
    """Per-frame (dx, dy); frame 0 (the reference frame) has zero shift."""
    d = np.zeros((n_frames, 2), dtype=np.float64)
    for n in range(1, n_frames):
        rng = np.random.Generator(np.random.PCG64(seed + n))
        d[n] = rng.uniform(-max_shift, max_shift, 2)
    return d


def noise_curves(alpha, beta, n=1001):
    """Analytic noise curves sigma_t(b), d_t(b) (float64[1001]).

    Un-clipped limits of the reference's Monte-Carlo (fast_monte_carlo.py:66-80):
    mean of ``np.std`` over 9 samples -> c4(9)*sqrt(8/9) = 0.9139 sigma, mean
    |difference of two 9-sample means| -> sqrt(2/pi)*sqrt(2/9) = 0.3761 sigma.
    The reference draws them with an unseeded RNG, so
    the curves are explicit inputs here.
    """
    b = np.linspace(0.0, 1.0, n)
    s = np.sqrt(alpha * b + beta)
    return 0.9139 * s, 0.3761 * s


def _render_np(H, W, dx, dy, cfa, params, dtype=np.float64):
    u, v, a, phi = params
    x = np.arange(W, dtype=np.float64) + dx
    y = np.arange(H, dtype=np.float64) + dy
    By = 2 * np.pi * np.outer(y, v)  # [H,K]
    cb, sb = np.cos(By), np.sin(By)
    chans = []
    for c in range(3):
        Ax = 2 * np.pi * np.outer(u, x) + phi[:, c : c + 1]  # [K,W]
        img = (cb * a) @ np.sin(Ax) + (sb * a) @ np.cos(Ax)
        chans.append(np.clip(0.5 + img, 0.02, 0.98))
    out = np.empty((H, W), dtype=np.float64)
    for i in range(2):
        for j in range(2):
            out[i::2, j::2] = chans[int(cfa[i][j])][i::2, j::2]
    return out


def make_burst(H, W, n_frames, seed=1234, alpha=ALPHA_ISO100, beta=BETA_ISO100,
               cfa=((0, 1), (1, 2)), wb=(1.0, 1.0, 1.0), max_shift=4.0, noise=True,
               occluder=False):
    """NumPy burst: returns (ref f32[H,W], comp f32[n-1,H,W], shifts f64[n,2]).

    ``wb`` multiplies the channels the way the reference's loader leaves them
    (utils_dng.py:153-160: raw is white-balanced, the robustness guide divides
    it back out).  ``occluder`` paints a moving bright square into the
    non-reference frames to exercise the robustness mask.
    """
    params = scene_params(seed)
    shifts = frame_shifts(n_frames, seed, max_shift)
    frames = np.empty((n_frames, H, W), dtype=np.float32)
    wbmap = np.empty((2, 2))
    for i in range(2):
        for j in range(2):
            wbmap[i, j] = wb[int(cfa[i][j])] / wb[1]
    for n in range(n_frames):
        clean = _render_np(H, W, shifts[n, 0], shifts[n, 1], cfa, params)
        if occluder and n > 0:
            s = max(8, min(H, W) // 8)
            y0 = (H // 3 + 5 * n) % max(1, H - s)
            x0 = (W // 2 + 7 * n) % max(1, W - s)
            clean[y0 : y0 + s, x0 : x0 + s] = 0.9
        if noise:
            rng = np.random.Generator(np.random.PCG64(seed + 1000 + n))
            clean = clean + np.sqrt(alpha * clean + beta) * rng.standard_normal((H, W))
        clean = np.clip(clean, 0.0, 1.0)
        for i in range(2):
            for j in range(2):
                clean[i::2, j::2] *= wbmap[i, j]
        frames[n] = clean.astype(np.float32)
    return frames[0], frames[1:], shifts


def make_burst_torch(H, W, n_frames, device, seed=1234, alpha=ALPHA_ISO100, beta=BETA_ISO100,
                     cfa=((0, 1), (1, 2)), max_shift=4.0, noise=True):
    """Same scene rendered with torch on ``device`` (float32 GEMMs; noise from
    torch's generator, so values differ from :func:`make_burst` at the noise
    level — use it for benchmarks, not for parity fixtures)."""
    import torch

    u, v, a, phi = scene_params(seed)
    shifts = frame_shifts(n_frames, seed, max_shift)
    f64 = torch.float64
    ut = torch.as_tensor(u, dtype=f64, device=device)
    vt = torch.as_tensor(v, dtype=f64, device=device)
    at = torch.as_tensor(a, dtype=f64, device=device)
    pht = torch.as_tensor(phi, dtype=f64, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    frames = torch.empty((n_frames, H, W), dtype=torch.float32, device=device)
    for n in range(n_frames):
        x = torch.arange(W, dtype=f64, device=device) + float(shifts[n, 0])
        y = torch.arange(H, dtype=f64, device=device) + float(shifts[n, 1])
        By = 2 * np.pi * torch.outer(y, vt)
        cb = (torch.cos(By) * at).float()
        sb = (torch.sin(By) * at).float()
        out = torch.empty((H, W), dtype=torch.float32, device=device)
        chans = {}
        for i in range(2):
            for j in range(2):
                c = int(cfa[i][j])
                if c not in chans:
                    Ax = 2 * np.pi * torch.outer(ut, x) + pht[:, c : c + 1]
                    img = cb @ torch.sin(Ax).float() + sb @ torch.cos(Ax).float()
                    chans[c] = torch.clamp(0.5 + img, 0.02, 0.98)
                out[i::2, j::2] = chans[c][i::2, j::2]
        if noise:
            out = out + torch.sqrt(alpha * out + beta) * torch.randn((H, W), generator=gen, device=device)
        frames[n] = torch.clamp(out, 0.0, 1.0)
        del chans
    return frames[0], frames[1:], shifts

"""Noise curves sigma_t(b), d_t(b) of the robustness noise model by Monte-Carlo (reference fast_monte_carlo.py;
SURVEY.md §8f-3) — seeded, on the GPU.

For brightness b in {0, 1/1000, ..., 1}: draw n_patches pairs of 3x3 patches b + sqrt(alpha b + beta) N(0,1)
clipped to [0, 1]; sigma_t = mean of the patches' standard deviation, d_t = mean |difference of the two patch
means| (fast_monte_carlo.py:44-84).  Like the reference, only the brightness levels where clipping matters are
simulated (within `tol` sigma of 0 or 1); in between sigma^2 and d^2 are linear in b and interpolated
(:126-157, :160-214).  The reference uses an unseeded NumPy RNG on all CPU cores (seconds); here one torch
generator on the device (milliseconds, reproducible)."""
import numpy as np
import torch

N_PATCHES = int(1e5)
N_BRIGHTNESS_LEVELS = 1000
TOL = 3


def get_non_linearity_bound(alpha, beta, tol=TOL):
    """Brightness range [xmin, xmax] inside which b +- tol sigma stays in [0, 1] (fast_monte_carlo.py:32-39)."""
    t2 = tol * tol
    xmin = t2 / 2 * (alpha + np.sqrt(t2 * alpha * alpha + 4 * beta))
    xmax = (2 + t2 * alpha - np.sqrt((2 + t2 * alpha) ** 2 - 4 * (1 + t2 * beta))) / 2
    return xmin, xmax


def regular_MC(b_array, alpha, beta, n_patches=N_PATCHES, seed=0, device=None, chunk=64):
    """(sigmas, diffs) float64 for every brightness of `b_array` (fast_monte_carlo.py:44-124)."""
    dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    b_all = torch.as_tensor(np.asarray(b_array, dtype=np.float64), device=dev)
    sig, dif = [], []
    for i in range(0, b_all.numel(), chunk):
        b = b_all[i:i + chunk].to(torch.float32)[:, None, None]            # [nb, 1, 1]
        std = torch.sqrt(b * float(alpha) + float(beta))
        shape = (b.shape[0], n_patches, 9)
        p1 = torch.clamp(b + std * torch.randn(shape, generator=gen, device=dev, dtype=torch.float32), 0.0, 1.0)
        p2 = torch.clamp(b + std * torch.randn(shape, generator=gen, device=dev, dtype=torch.float32), 0.0, 1.0)
        s = 0.5 * (p1.std(dim=2, unbiased=False) + p2.std(dim=2, unbiased=False)).double().mean(dim=1)
        d = (p1.mean(dim=2) - p2.mean(dim=2)).abs().double().mean(dim=1)
        sig.append(s)
        dif.append(d)
    return torch.cat(sig).cpu().numpy(), torch.cat(dif).cpu().numpy()


def interp_MC(b_array, sigma_min, sigma_max, diff_min, diff_max):
    """Linear interpolation of sigma^2 and d^2 between the two ends (fast_monte_carlo.py:126-157)."""
    nb = (b_array - b_array[0]) / (b_array[-1] - b_array[0])
    s2 = nb * (sigma_max ** 2 - sigma_min ** 2) + sigma_min ** 2
    d2 = nb * (diff_max ** 2 - diff_min ** 2) + diff_min ** 2
    return np.sqrt(s2[1:-1]), np.sqrt(d2[1:-1])


def run_fast_MC(alpha, beta, seed=0, device=None, n_patches=N_PATCHES):
    """(std_curve, diff_curve): float64[1001] for brightness 0, 0.001, ..., 1 (fast_monte_carlo.py:160-214)."""
    n = N_BRIGHTNESS_LEVELS
    xmin, xmax = get_non_linearity_bound(alpha, beta, TOL)
    imin = int(np.ceil(xmin * n)) + 1
    imax = int(np.floor(xmax * n)) - 1
    brightness = np.arange(n + 1) / n
    if imin > n or imax <= imin:
        return regular_MC(brightness, alpha, beta, n_patches, seed, device)
    sigmas, diffs = np.empty(n + 1), np.empty(n + 1)
    nl = np.concatenate((brightness[:imin + 1], brightness[imax:]))
    s_nl, d_nl = regular_MC(nl, alpha, beta, n_patches, seed, device)
    sigmas[:imin + 1], diffs[:imin + 1] = s_nl[:imin + 1], d_nl[:imin + 1]
    sigmas[imax:], diffs[imax:] = s_nl[imin + 1:], d_nl[imin + 1:]
    s_l, d_l = interp_MC(brightness[imin - 1:imax + 2], sigmas[imin], sigmas[imax], diffs[imin], diffs[imax])
    sigmas[imin:imax + 1] = s_l
    diffs[imin:imax + 1] = d_l
    return sigmas, diffs

"""CPU restatement of the burst front end (test infrastructure only, see oracle/__init__.py).

normalize_burst : reference utils_dng.py:149-160 (float32 array arithmetic with Python scalars).
unitary_mc / run_fast_mc : reference fast_monte_carlo.py:44-84, 126-214 (NumPy, seeded generator instead of the
global unseeded one).  Pinning: the normalisation against what the reference's own load_dng_burst returned for three
synthetic sensors (golden `frontend`, tools/refsim/make_goldens.py::stage_frontend: the .dng DECODER is a stand-in —
rawpy is absent — the loader is upstream code) and against hand-computed values (tests/test_host_logic.py); the
Monte-Carlo only statistically (the reference's own draws are unseeded, SURVEY.md
App. A D18) — against the analytic un-clipped limits and against the product's GPU estimator.
"""
import numpy as np

F32 = np.float32


def normalize_burst(raw, black_levels, white_level, white_balance, cfa):
    """integer counts [n, H, W] -> float32, per CFA colour (utils_dng.py:149-160)."""
    out = np.asarray(raw).astype(F32)
    cfa = np.asarray(cfa)
    for i in range(2):
        for j in range(2):
            c = int(cfa[i, j])
            k = white_balance[c] / white_balance[1]
            out[..., i::2, j::2] = (out[..., i::2, j::2] - black_levels[c]) / (white_level - black_levels[c])
            out[..., i::2, j::2] *= k
    return out


def unitary_mc(alpha, beta, b, n_patches, rng):
    """(diff_mean, std_mean) at brightness b (fast_monte_carlo.py:44-84)."""
    patch = np.ones((n_patches, 3, 3)) * b
    s = np.sqrt(patch * alpha + beta)
    p1 = np.clip(patch + s * rng.standard_normal(patch.shape), 0.0, 1.0)
    p2 = np.clip(patch + s * rng.standard_normal(patch.shape), 0.0, 1.0)
    std_mean = 0.5 * np.mean(np.std(p1, axis=(1, 2)) + np.std(p2, axis=(1, 2)))
    diff_mean = np.mean(np.abs(np.mean(p1, axis=(1, 2)) - np.mean(p2, axis=(1, 2))))
    return diff_mean, std_mean


def non_linearity_bound(alpha, beta, tol=3):
    t2 = tol * tol
    xmin = t2 / 2 * (alpha + np.sqrt(t2 * alpha * alpha + 4 * beta))
    xmax = (2 + t2 * alpha - np.sqrt((2 + t2 * alpha) ** 2 - 4 * (1 + t2 * beta))) / 2
    return xmin, xmax

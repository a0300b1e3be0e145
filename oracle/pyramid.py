"""Oracle: Gaussian pyramid (restates reference utils_image.py:360-391, alignment.py:74-82).  Test infrastructure."""
import numpy as np

F32 = np.float32


def gaussian_taps(factor):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma=f/2, order=0, radius=int(2f+0.5)),
    as called at reference utils_image.py:380, restated from its definition."""
    sigma = factor * 0.5
    radius = int(4 * factor * 0.5 + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x.astype(np.float64) ** 2)
    return (phi / phi.sum()).astype(F32), radius


def downsample(img, factor):
    """Valid separable Gaussian (rows first, then columns) in float32, then ``[::f]`` over
    ``floor(size/f)*f`` samples (reference utils_image.py:380-391).  factor 1 = identity."""
    img = np.asarray(img, dtype=F32)
    if factor == 1:
        return img
    g, r = gaussian_taps(factor)
    H, W = img.shape
    Hf, Wf = H - 2 * r, W - 2 * r
    tmp = np.zeros((Hf, W), dtype=F32)
    for i in range(2 * r + 1):  # convolve y
        tmp += g[i] * img[i : i + Hf, :]
    out = np.zeros((Hf, Wf), dtype=F32)
    for j in range(2 * r + 1):  # convolve x
        out += g[j] * tmp[:, j : j + Wf]
    h2, w2 = Hf // factor, Wf // factor
    return np.ascontiguousarray(out[: h2 * factor : factor, : w2 * factor : factor])


def build_gaussian_pyramid(img, factors):
    """reference alignment.py:74-82; returned COARSE FIRST like the reference."""
    pyr = [downsample(img, factors[0])]
    for f in factors[1:]:
        pyr.append(downsample(pyr[-1], f))
    return pyr[::-1]

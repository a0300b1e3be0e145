"""Oracle: raw -> grey image (restates reference utils_image.py:58-112, 346-357).  Test infrastructure."""
import numpy as np

F32 = np.float32


def grey_fft(img):
    """Alg. 3, ideal half-band low-pass (reference utils_image.py:82-100).

    Full complex FFT, fftshift, zero the outer quarter bands, inverse, keep the
    real part.  ``[-H//4:]`` parses as ``(-H)//4`` = -ceil(H/4), so the kept
    band is bins -ceil(H/4)+... : the mask is asymmetric by one bin for H%4==0.
    Computed here in float64 (the reference runs cuFFT in complex64).
    """
    img = np.asarray(img, dtype=F32)
    H, W = img.shape
    f = np.fft.fftshift(np.fft.fft2(img.astype(np.float64)))
    f[: H // 4, :] = 0
    f[:, : W // 4] = 0
    f[(-H) // 4 :, :] = 0
    f[:, (-W) // 4 :] = 0
    out = np.fft.ifft2(np.fft.ifftshift(f))
    return out.real.astype(F32)


def decimate_to_grey(img):
    """2x2 mean (reference utils_image.py:346-357).  The accumulator ``c`` starts as
    an int and is unified to float64 by Numba, so the sum is exact in float64."""
    img = np.asarray(img, dtype=F32)
    h, w = img.shape[0] // 2, img.shape[1] // 2
    v = img[: 2 * h, : 2 * w].astype(np.float64)
    c = v[0::2, 0::2] + v[0::2, 1::2]
    c = c + v[1::2, 0::2]
    c = c + v[1::2, 1::2]
    return (c / 4).astype(F32)


def compute_grey_images(img, method):
    """reference utils_image.py:58-112."""
    if method == "FFT":
        return grey_fft(img)
    if method == "decimating":
        return decimate_to_grey(img)
    raise NotImplementedError(method)

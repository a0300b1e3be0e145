// The step after the hot path, on the device: frame-count denoisers driven by the accumulated
// robustness (reference utils_image.py:174-309), raw2rgb.postprocess — colour matrix, unsharp mask, devignetting, gamma
// (raw2rgb.py:113-160, 198-250) — and the EXIF orientation (utils_image.py:12-55) folded into the final store.
// The reference runs all of it on the host (NumPy / skimage) after copying the 48 MP image back; here the image never
// leaves HBM until it is finished: 2 passes over [sH][sW][3] (vertical blur -> horizontal blur + everything else).
#include "hhsr_common.h"
#include <math.h>

// ---- accumulated-robustness sample an output pixel reads (utils_image.py:205-206, 262-263) ---------------------------
// half_index = 1: int(round((y - 0.5) / (2 scale))), the reference's expression (it addresses the [H][W] map as if it
// had half that resolution — a deterministic upstream quirk, reproduced by default); 0: the nearest raw pixel;
// 2: int(round(y / scale)), the reference's `mode: grey` branch (utils_image.py:203-204, 260-261).
__device__ __forceinline__ int acc_index(int i, double scale, int half_index, int n) {
    const double v = half_index == 2 ? (double)i / scale
                   : half_index ? ((double)i - 0.5) / (2.0 * scale) : ((double)i + 0.5) / scale - 0.5;
    return min(max((int)rint(v), 0), n - 1);  // rint: half to even, like Python's round()
}

// ---- median denoiser ------------------------------------------------------------------------------------------------------
// radius = round(radius_max (mfc - min(r, mfc)) / mfc) <= 7; out = element k // 2 of the k in-image samples of the
// (2 radius + 1)^2 window in ascending order (the reference bubble-sorts a 256-element local array).
__global__ void __launch_bounds__(256) k_fc_median(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                    const float* __restrict__ racc, int ah, int aw, double scale,
                                                    double radius_max, double mfc, int half_index) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), c = blockIdx.z;
    if (x >= W || y >= H) return;
    const double r = fmin((double)racc[(size_t)acc_index(y, scale, half_index, ah) * aw + acc_index(x, scale, half_index, aw)], mfc);
    const int radius = min(7, (int)rint(radius_max * (mfc - r) / mfc));  // host rejects radius_max > 7
    const int y0 = max(y - radius, 0), y1 = min(y + radius, H - 1), x0 = max(x - radius, 0), x1 = min(x + radius, W - 1);
    const int k = (y1 - y0 + 1) * (x1 - x0 + 1), want = k / 2;
    // the window goes to a private buffer (the reference's cuda.local.array), then rank selection: the answer is the
    // sample v with #(< v) <= want < #(<= v) — element `want` of the ascending order without sorting
    float buf[225];
    int n = 0, nans = 0;
    for (int yi = y0; yi <= y1; ++yi)
        for (int xi = x0; xi <= x1; ++xi) {
            const float v = in[((size_t)yi * W + xi) * 3 + c];
            nans += v != v;
            buf[n++] = v;
        }
    float res = buf[0];
    if (nans == 0) {
        for (int i = 0; i < k; ++i) {
            const float v = buf[i];
            int less = 0, leq = 0;
            for (int j = 0; j < k; ++j) {
                less += buf[j] < v;
                leq += buf[j] <= v;
            }
            if (less <= want && want < leq) {
                res = v;
                break;
            }
        }
    } else {
        // a NaN sample (a pixel without any sample of a colour, D6) is not ordered; the reference's bubble sort
        // (utils_image.py:296-302) then leaves a specific un-sorted arrangement: replay it literally
        for (int i = 0; i < k - 1; ++i)
            for (int j = 0; j < k - i - 1; ++j)
                if (buf[j] > buf[j + 1]) {
                    const float t = buf[j];
                    buf[j] = buf[j + 1];
                    buf[j + 1] = t;
                }
        res = buf[want];
    }
    out[((size_t)y * W + x) * 3 + c] = res;
}

// ---- gauss denoiser --------------------------------------------------------------------------------------------------------
// sigma = sigma_max (mfc - min(r, mfc)) / mfc, window |i|, |j| <= t = ceil(3 sigma) (the reference's range(-t, t + 1) of
// a float does not type under Numba: the build defines the window), w = exp(-(i^2 + j^2) / (2 sigma^2)), float64 sums.
__global__ void __launch_bounds__(256) k_fc_gauss(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                   const float* __restrict__ racc, int ah, int aw, double scale,
                                                   double sigma_max, double mfc, int half_index) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), c = blockIdx.z;
    if (x >= W || y >= H) return;
    const double r = fmin((double)racc[(size_t)acc_index(y, scale, half_index, ah) * aw + acc_index(x, scale, half_index, aw)], mfc);
    const double sigma = sigma_max * (mfc - r) / mfc;
    const int t = (int)ceil(3.0 * sigma);
    double num = 0.0, den = 0.0;
    for (int i = -t; i <= t; ++i)
        for (int j = -t; j <= t; ++j) {
            const int xi = x + j, yi = y + i;
            if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;
            const double w = sigma == 0.0 ? (double)(i == 0 && j == 0) : exp(-(double)(j * j + i * i) / (2.0 * sigma * sigma));
            num += w * (double)in[((size_t)yi * W + xi) * 3 + c];
            den += w;
        }
    out[((size_t)y * W + x) * 3 + c] = (float)(num / den);
}

// ---- postprocess -----------------------------------------------------------------------------------------------------------
// np.clip keeps NaN (the merged image has NaN pixels where a colour has no sample at all, D6); fminf / fmaxf would not
__device__ __forceinline__ float clip01(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
__device__ __forceinline__ double clip01d(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

struct PostArgs {
    float ccm[9];       // cam2rgb, row major (do_ccm)
    int do_ccm, do_sharpen, do_devignette, do_gamma;
    float amount;       // unsharp amount
    int radius;         // blur taps: -radius .. radius
    int H, W;           // image
    int orientation;    // EXIF 1..8
};

__device__ __forceinline__ void load_ccm(const float* __restrict__ img, size_t o, const PostArgs& a, float v[3]) {
    const float r = img[o], g = img[o + 1], b = img[o + 2];
    if (a.do_ccm) {  // np.matmul(ccm, image) then clip to [0, 1] (raw2rgb.py:131-138, 221-222)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s = a.ccm[3 * k] * r + a.ccm[3 * k + 1] * g + a.ccm[3 * k + 2] * b;
            v[k] = clip01(s);
        }
    } else {
        v[0] = r; v[1] = g; v[2] = b;
    }
}

__device__ __forceinline__ int reflect(int i, int n) {  // scipy.ndimage mode "reflect": (d c b a | a b c d | d c b a)
    const int p = 2 * n;
    i = ((i % p) + p) % p;
    return i < n ? i : p - 1 - i;
}

// pass 1: Gaussian along y (axis 0 first, like scipy.ndimage.gaussian_filter) of the colour-corrected image; float64
// accumulation, float32 result (scipy's intermediate has the input's dtype).  64 x 4 pixels per workgroup.
__global__ void __launch_bounds__(256) k_post_vblur(const float* __restrict__ img, float* __restrict__ tmp, PostArgs a,
                                                     const double* __restrict__ taps) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int i = -a.radius; i <= a.radius; ++i) {
        float v[3];
        load_ccm(img, ((size_t)reflect(y + i, a.H) * a.W + x) * 3, a, v);
        const double w = taps[i + a.radius];
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += w * (double)v[k];
    }
    const size_t o = ((size_t)y * a.W + x) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) tmp[o + k] = (float)acc[k];
}

// orientation: where output pixel (y, x) of the un-oriented image goes (utils_image.py:12-55); returns the flat index
// and leaves the oriented shape to the host
__device__ __forceinline__ size_t oriented_index(int y, int x, int H, int W, int ori) {
    switch (ori) {
        case 2: return (size_t)y * W + (W - 1 - x);                 // mirror horizontal
        case 3: return (size_t)(H - 1 - y) * W + (W - 1 - x);       // rotate 180
        case 4: return (size_t)(H - 1 - y) * W + x;                 // mirror vertical
        case 5: return (size_t)x * H + y;                           // transpose            -> [W][H]
        case 6: return (size_t)x * H + (H - 1 - y);                 // rotate 90 CW         -> [W][H]
        case 7: return (size_t)(W - 1 - x) * H + (H - 1 - y);       // anti-transpose       -> [W][H]
        case 8: return (size_t)(W - 1 - x) * H + y;                 // rotate 270 CW        -> [W][H]
        default: return (size_t)y * W + x;
    }
}

// pass 2: Gaussian along x of pass 1 (sharpening), result = c + (c - blurred) amount, devignetting, clip, gamma, clip,
// store at the oriented position.  Without sharpening this is the only pass.
__global__ void __launch_bounds__(256) k_post_finish(const float* __restrict__ img, const float* __restrict__ tmp,
                                                      float* __restrict__ out, PostArgs a, const double* __restrict__ taps) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.W || y >= a.H) return;
    float c[3];
    load_ccm(img, ((size_t)y * a.W + x) * 3, a, c);
    if (a.do_sharpen) {
        double acc[3] = {0.0, 0.0, 0.0};
        for (int j = -a.radius; j <= a.radius; ++j) {
            const size_t o = ((size_t)y * a.W + reflect(x + j, a.W)) * 3;
            const double w = taps[j + a.radius];
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[k] += w * (double)tmp[o + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = c[k] + (c[k] - (float)acc[k]) * a.amount;  // float32, no clipping (preserve_range)
    }
    double d[3] = {(double)c[0], (double)c[1], (double)c[2]};
    if (a.do_devignette) {  // (2 - cos(|ly| |lx|)^4) with ly in +- h/w pi/2, lx in +- pi/2 (raw2rgb.py:198-204); float64
        const double hw = (double)a.H / (double)a.W * 1.57079632679489661923;
        const double ly = a.H > 1 ? fabs(-hw + 2.0 * hw * (double)y / (double)(a.H - 1)) : hw;
        const double lx = a.W > 1 ? fabs(-1.57079632679489661923 + 3.14159265358979323846 * (double)x / (double)(a.W - 1)) : 1.57079632679489661923;
        const double cv = cos(ly * lx);
        const double gain = 2.0 - cv * cv * cv * cv;
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k] *= gain;
    }
    const size_t o = oriented_index(y, x, a.H, a.W, a.orientation) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v;
        if (a.do_devignette) {  // float64 from here on upstream (float64 gain x float32 image)
            double t = clip01d(d[k]);
            if (a.do_gamma) t = pow(t, 1.0 / 2.2);
            v = (float)clip01d(t);
        } else {
            float t = clip01(c[k]);
            if (a.do_gamma) t = powf(t, (float)(1.0 / 2.2));
            v = clip01(t);
        }
        out[o + k] = v;
    }
}

// [H][W] map (accumulated robustness) to its oriented position
__global__ void __launch_bounds__(256) k_orient_plane(const float* __restrict__ in, float* __restrict__ out, int H, int W,
                                                       int ori) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    out[oriented_index(y, x, H, W, ori)] = in[(size_t)y * W + x];
}

extern "C" int hhsr_frame_count_denoise(const float* image, float* out, int H, int W, const float* acc_r, int ah, int aw,
                                        double scale, int kind, double strength_max, double max_frame_count,
                                        int half_index, void* stream) {
    HHSR_ARG(image && out && acc_r && image != out);
    HHSR_ARG(H > 0 && W > 0 && ah > 0 && aw > 0 && scale >= 1.0 && max_frame_count > 0.0 && strength_max >= 0.0);
    HHSR_ARG(kind == 0 || kind == 1);
    const dim3 grid(hhsr_cdiv(W, 64), hhsr_cdiv(H, 4), 3), block(256);
    if (kind == 0) {
        if (rint(strength_max) > 7.0) {
            hhsr_set_error("hhsr_frame_count_denoise: median radius_max %g > 7 overflows the reference's 16 x 16 sample "
                           "buffer (utils_image.py:272): undefined upstream, rejected here", strength_max);
            return -2;
        }
        hipLaunchKernelGGL(k_fc_median, grid, block, 0, (hipStream_t)stream, image, out, H, W, acc_r, ah, aw, scale,
                           strength_max, max_frame_count, half_index);
    } else {
        hipLaunchKernelGGL(k_fc_gauss, grid, block, 0, (hipStream_t)stream, image, out, H, W, acc_r, ah, aw, scale,
                           strength_max, max_frame_count, half_index);
    }
    HHSR_LAUNCHED();
}

extern "C" int hhsr_postprocess(const float* image, float* tmp, float* out, int H, int W, const float* cam2rgb,
                                int do_sharpen, double amount, const double* taps, int radius, int do_devignette,
                                int do_gamma, int orientation, void* stream) {
    HHSR_ARG(image && out && image != out && H > 0 && W > 0);
    HHSR_ARG(orientation >= 1 && orientation <= 8);
    HHSR_ARG(!do_sharpen || (tmp && taps && radius >= 0 && radius <= 64 && tmp != image && tmp != out));
    PostArgs a;
    a.do_ccm = cam2rgb != nullptr;
    for (int k = 0; k < 9; ++k) a.ccm[k] = cam2rgb ? cam2rgb[k] : (k % 4 == 0 ? 1.f : 0.f);
    a.do_sharpen = do_sharpen != 0;
    a.do_devignette = do_devignette != 0;
    a.do_gamma = do_gamma != 0;
    a.amount = (float)amount;
    a.radius = radius;
    a.H = H;
    a.W = W;
    a.orientation = orientation;
    const dim3 grid(hhsr_cdiv(W, 64), hhsr_cdiv(H, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (a.do_sharpen) hipLaunchKernelGGL(k_post_vblur, grid, block, 0, s, image, tmp, a, taps);
    hipLaunchKernelGGL(k_post_finish, grid, block, 0, s, image, tmp, out, a, taps);
    HHSR_LAUNCHED();
}

extern "C" int hhsr_orient_plane(const float* in, float* out, int H, int W, int orientation, void* stream) {
    HHSR_ARG(in && out && in != out && H > 0 && W > 0 && orientation >= 1 && orientation <= 8);
    hipLaunchKernelGGL(k_orient_plane, dim3(hhsr_cdiv(W, 64), hhsr_cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream, in, out,
                       H, W, orientation);
    HHSR_LAUNCHED();
}

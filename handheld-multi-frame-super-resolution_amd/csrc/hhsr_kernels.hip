// Alg. 5 kernel covariances, fully fused: GAT -> 2x2 mean -> 2x2 gradients -> 2x2-window structure
// tensor -> eigen-decomposition -> (k1, k2) -> covariance, one thread per Bayer quad
// (reference kernels.py:29-243, utils_image.py:117-170 and 346-357, linalg.py:87-185).
//
// Bytes: 4 B/raw pixel in, 4 B/raw pixel out (16 B per quad): HBM bound.  The variance-stabilised
// quad means are staged in LDS with a one-quad halo so each raw pixel is transformed ~1.27x.
// Arithmetic follows the reference's Numba typing: the GAT and the eigenvalue
// roots are float64 (MI355X runs fp64 at half the fp32 vector rate, so this is affordable), storage
// and the tensor sums are float32.
#include "hhsr_common.h"

constexpr int CV_TX = 32, CV_TY = 8;  // quads per 256-thread workgroup

struct CovParams {
    double alpha, beta, k_detail, k_denoise, D_th, D_tr, k_stretch, k_shrink;
    int law;
};

__device__ __forceinline__ float gat1(float v, double alpha, double c0, double two_over_alpha) {
    // VST = alpha*I + 3/8*alpha^2 + beta ; max(0, .) ; 2/alpha * sqrt(.)   (utils_image.py:167-170)
    double t = alpha * (double)v + c0;
    t = t > 0.0 ? t : 0.0;
    return (float)(two_over_alpha * sqrt(t));
}

__global__ void __launch_bounds__(256) k_cov_from_raw(const float* __restrict__ raw, int H, int W, int pitch,
                                                       float4* __restrict__ covs, int gh, int gw, CovParams P) {
    // grey tile with a halo of one quad on the top/left and one on the bottom/right
    __shared__ float s_g[CV_TY + 2][CV_TX + 2 + 1];
    const int qx0 = blockIdx.x * CV_TX, qy0 = blockIdx.y * CV_TY;
    const double c0 = 3.0 / 8.0 * P.alpha * P.alpha + P.beta;
    const double toa = 2.0 / P.alpha;
    for (int p = threadIdx.x; p < (CV_TY + 2) * (CV_TX + 2); p += 256) {
        const int i = p / (CV_TX + 2), j = p - i * (CV_TX + 2);
        const int qy = qy0 + i - 1, qx = qx0 + j - 1;
        float g = 0.f;
        if (qy >= 0 && qy < gh && qx >= 0 && qx < gw) {
            const float2 a = *reinterpret_cast<const float2*>(raw + (size_t)(2 * qy) * pitch + 2 * qx);
            const float2 b = *reinterpret_cast<const float2*>(raw + (size_t)(2 * qy + 1) * pitch + 2 * qx);
            // decimate: float64 sum of the four float32 VST values, /4 (utils_image.py:346-357)
            const double s = (double)gat1(a.x, P.alpha, c0, toa) + (double)gat1(a.y, P.alpha, c0, toa) +
                             (double)gat1(b.x, P.alpha, c0, toa) + (double)gat1(b.y, P.alpha, c0, toa);
            g = (float)(s / 4.0);
        }
        s_g[i][j] = g;
    }
    __syncthreads();
    const int lx = threadIdx.x % CV_TX, ly = threadIdx.x / CV_TX;
    const int qx = qx0 + lx, qy = qy0 + ly;
    if (qx >= gw || qy >= gh) return;
    // structure tensor over the gradient samples (y-1..y, x-1..x) that exist in the [gh-1][gw-1] grid
    float T00 = 0.f, T01 = 0.f, T11 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy_ = qy - 1 + i, gx_ = qx - 1 + j;
            if (gy_ >= 0 && gy_ < gh - 1 && gx_ >= 0 && gx_ < gw - 1) {
                const float g00 = s_g[ly + i][lx + j], g01 = s_g[ly + i][lx + j + 1];
                const float g10 = s_g[ly + i + 1][lx + j], g11 = s_g[ly + i + 1][lx + j + 1];
                // two chained float32 convs (kernels.py:97-116)
                const float t0a = -0.5f * g00 + 0.5f * g01, t0b = -0.5f * g10 + 0.5f * g11;
                const float t1a = 0.5f * g00 + 0.5f * g01, t1b = 0.5f * g10 + 0.5f * g11;
                const float vx = 0.5f * t0a + 0.5f * t0b;
                const float vy = -0.5f * t1a + 0.5f * t1b;
                T00 += vx * vx;
                T01 += vx * vy;
                T11 += vy * vy;
            }
        }
    }
    // eigenvalues (linalg.py:87-130): float32 b, c; float64 discriminant and roots; stored float32
    const float b = -(T00 + T11);
    const float c = T00 * T11 - T01 * T01;
    const float bb = b * b;
    double delta = (double)bb - 4.0 * (double)c;
    delta = delta > 0.0 ? delta : 0.0;
    const double sq = sqrt(delta);
    const double r1 = (-(double)b + sq) / 2.0, r2 = (-(double)b - sq) / 2.0;
    float l1, l2;
    if (fabs(r1) >= fabs(r2)) {
        l1 = (float)r1;
        l2 = (float)r2;
    } else {
        l1 = (float)r2;
        l2 = (float)r1;
    }
    // eigenvectors (linalg.py:133-179)
    float e1x, e1y, e2x, e2y;
    if (T01 == 0.f && T00 == T11) {
        e1x = 1.f; e1y = 0.f; e2x = 0.f; e2y = 1.f;
    } else {
        const float a0 = T00 + T01 - l2, a1 = T01 + T11 - l2;
        if (a0 == 0.f) {
            e1x = a0; e1y = 1.f; e2x = 1.f; e2y = 0.f;
        } else if (a1 == 0.f) {
            e1x = 1.f; e1y = a1; e2x = 0.f; e2y = 1.f;
        } else {
            const float nrm = sqrtf(a0 * a0 + a1 * a1);
            e1x = a0 / nrm;
            e1y = a1 / nrm;
            const float sgn = copysignf(1.f, e1x);
            e2y = fabsf(e1x);
            e2x = -e1y * sgn;
        }
    }
    // k1, k2 (kernels.py:195-243): A, D float64 from float32 square roots; k stored float32
    const double A = 1.0 + (double)sqrtf((l1 - l2) / (l1 + l2));
    double D = 1.0 - (double)sqrtf(l1) / P.D_tr + P.D_th;
    D = D > 0.0 ? D : 0.0;  // clamp with Python max/min semantics (NaN -> 0)
    D = D < 1.0 ? D : 1.0;
    double k1d, k2d;
    if (P.law == 0) {  // hard_threshold; a NaN anisotropy falls into the else branch
        if (A > 1.95) {
            k1d = 1.0 / P.k_shrink;
            k2d = P.k_stretch;
        } else {
            k1d = 1.0;
            k2d = 1.0;
        }
    } else {  // linear
        k1d = 1.0 + A / 2.0 * (1.0 / P.k_shrink - 1.0);
        k2d = 1.0 + A / 2.0 * (P.k_stretch - 1.0);
    }
    const float k1 = (float)(P.k_detail * ((1.0 - D) * k1d + D * P.k_denoise));
    const float k2 = (float)(P.k_detail * ((1.0 - D) * k2d + D * P.k_denoise));
    const float k1s = k1 * k1, k2s = k2 * k2;
    float4 o;
    o.x = k1s * e1x * e1x + k2s * e2x * e2x;
    o.y = k1s * e1x * e1y + k2s * e2x * e2y;
    o.z = o.y;
    o.w = k1s * e1y * e1y + k2s * e2y * e2y;
    covs[(size_t)qy * gw + qx] = o;
}

extern "C" int hhsr_cov_from_raw(const float* raw, int H, int W, int pitch, float* covs, double alpha, double beta,
                                 double k_detail, double k_denoise, double D_th, double D_tr, double k_stretch,
                                 double k_shrink, int law, void* stream) {
    HHSR_ARG(raw && covs && H >= 2 && W >= 2 && pitch >= W);
    HHSR_ARG((pitch & 1) == 0 && ((uintptr_t)raw & 7) == 0 && ((uintptr_t)covs & 15) == 0);
    HHSR_ARG(alpha > 0.0);  // utils_image.py:141: the VST is ill-defined otherwise
    HHSR_ARG(law == 0 || law == 1);
    const int gh = H / 2, gw = W / 2;
    CovParams P{alpha, beta, k_detail, k_denoise, D_th, D_tr, k_stretch, k_shrink, law};
    hipLaunchKernelGGL(k_cov_from_raw, dim3(hhsr_cdiv(gw, CV_TX), hhsr_cdiv(gh, CV_TY)), dim3(256), 0,
                       (hipStream_t)stream, raw, H, W, pitch, reinterpret_cast<float4*>(covs), gh, gw, P);
    HHSR_LAUNCHED();
}

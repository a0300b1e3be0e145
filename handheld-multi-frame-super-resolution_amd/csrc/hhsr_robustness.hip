// Alg. 6-9 robustness (reference robustness.py; Dodgson kernel utils_image.py:395-406).
//
// The reference runs 8 launches and ~600 MB of raw-resolution temporaries per frame.  Here:
//   k_frame_stats raw -> guide (LDS tile) -> 3x3 mean/variance at guide resolution        (hhsr_kernels.hip)
//   k_rob_frame   fused: Dodgson warp-upsample of the frame's means, |d mu|, noise-model shrink,
//                 S lookup, threshold -> R at raw resolution                               (no temporaries)
//   k_local_min5  5x5 minimum through an LDS tile
// Arithmetic follows the reference's Numba typing: float64 weights and noise-model
// maths, float32 storage and float32 running sums that are rounded after every tap.
#include "hhsr_common.h"
#include <stdlib.h>

// (the guide image + local statistics pass lives in hhsr_kernels.hip: it shares its raw tile with the kernel
// covariances, k_frame_stats)
// ---- Dodgson quadratic warp-upsample ------------------------------------------------------------
__device__ __forceinline__ double dodgson(double x) {  // utils_image.py:399-406
    const double a = fabs(x);
    if (a <= 0.5) return -2.0 * a * a + 1.0;
    if (a <= 1.5) return a * a - 5.0 / 2.0 * a + 1.5;
    return 0.0;
}

__device__ __forceinline__ float dodgsonf(float x) {
    const float a = fabsf(x);
    if (a <= 0.5f) return fmaf(-2.0f * a, a, 1.0f);
    if (a <= 1.5f) return fmaf(a, a, fmaf(-2.5f, a, 1.5f));
    return 0.0f;
}

// Interpolates the NC channels of a [NC][lh][lw] map at raw pixel (y, x) displaced by (fx, fy).
// Returns false (outside the guide image -> +inf, robustness.py:386-391) or true with out[NC].
template <int NC>
__device__ __forceinline__ bool dodgson_sample(const float* __restrict__ LR, int lh, int lw, int y, int x, double fx,
                                               double fy, float out[NC]) {
    const double ly = ((double)y + fy + 0.5) / 2.0 - 0.5;
    const double lx = ((double)x + fx + 0.5) / 2.0 - 0.5;
    if (!(ly >= 0.0 && ly < (double)lh && lx >= 0.0 && lx < (double)lw)) return false;
    const int cy = (int)rint(ly), cx = (int)rint(lx);  // round-half-even
    const size_t plane = (size_t)lh * lw;
    float b[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) b[c] = 0.f;
    double wacc = 0.0;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
        const int y_ = clampi(cy + i, 0, lh - 1);
        const double wy = dodgson((double)y_ - ly);
#pragma unroll
        for (int j = -1; j <= 1; ++j) {
            const int x_ = clampi(cx + j, 0, lw - 1);
            const double w = wy * dodgson((double)x_ - lx);
            const size_t o = (size_t)y_ * lw + x_;
            // float32 buffer += float32 * float64, rounded after every tap (robustness.py:414-415)
#pragma unroll
            for (int c = 0; c < NC; ++c) b[c] = (float)((double)b[c] + (double)LR[c * plane + o] * w);
            wacc += w;
        }
    }
    // interior, un-warped or not: the Dodgson weights sum to exactly 1 there and x / 1.0 == x
#pragma unroll
    for (int c = 0; c < NC; ++c) out[c] = wacc == 1.0 ? b[c] : (float)((double)b[c] / wacc);
    return true;
}

__global__ void __launch_bounds__(256) k_rob_upscale(const float* __restrict__ LR, int lh, int lw,
                                                      const float2* __restrict__ flow, int nx, int ts,
                                                      float* __restrict__ HR, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    double fx = 0.0, fy = 0.0;
    if (flow) {
        const float2 f = flow[(size_t)(y / ts) * nx + x / ts];
        fx = (double)f.x;
        fy = (double)f.y;
    }
    float v[3];
    const size_t plane = (size_t)H * W, o = (size_t)y * W + x;
    if (!dodgson_sample<3>(LR, lh, lw, y, x, fx, fy, v)) v[0] = v[1] = v[2] = INFINITY;
    HR[o] = v[0];
    HR[plane + o] = v[1];
    HR[2 * plane + o] = v[2];
}

extern "C" int hhsr_rob_upscale(const float* stats, int lh, int lw, const float* flow, int ny, int nx, int ts,
                                float* out, void* stream) {
    HHSR_ARG(stats && out && lh > 0 && lw > 0);
    const int H = 2 * lh, W = 2 * lw;
    if (flow) HHSR_ARG(ts > 0 && ny * ts >= H && nx * ts >= W);
    hipLaunchKernelGGL(k_rob_upscale, dim3(hhsr_cdiv(W, 64), hhsr_cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream,
                       stats, lh, lw, reinterpret_cast<const float2*>(flow), nx, ts > 0 ? ts : 1, out, H, W);
    HHSR_LAUNCHED();
}

// ---- flow irregularity S (robustness.py:570-612) ------------------------------------------------
// rows_lo / rows_hi: tile rows that lie in memory before flow[0] / after flow[ny - 1] — `flow` is then a row slice of a
// larger field (the row slabs of the multi-GPU path) and the 3 x 3 neighbourhood reads them, so that the weights of the
// slice's first and last rows are those of the full field.
__global__ void __launch_bounds__(256) k_rob_s(const float2* __restrict__ flow, int ny, int nx, double Mt2, float s1,
                                                float s2, float* __restrict__ S, int rows_lo, int rows_hi) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j) {
            const int yy = y + i, xx = x + j;
            if (yy >= -rows_lo && yy < ny + rows_hi && xx >= 0 && xx < nx) {
                const float2 f = flow[(ptrdiff_t)yy * nx + xx];
                mxx = fmaxf(mxx, f.x); mxy = fmaxf(mxy, f.y);
                mnx = fminf(mnx, f.x); mny = fminf(mny, f.y);
            }
        }
    const float d0 = mxx - mnx, d1 = mxy - mny;
    const float m = d0 * d0 + d1 * d1;
    S[(size_t)y * nx + x] = ((double)m > Mt2) ? s1 : s2;
}

extern "C" int hhsr_rob_s(const float* flow, int ny, int nx, double Mt, float s1, float s2, float* S, int rows_before,
                          int rows_after, void* stream) {
    HHSR_ARG(flow && S && ny > 0 && nx > 0 && rows_before >= 0 && rows_after >= 0);
    hipLaunchKernelGGL(k_rob_s, dim3(hhsr_cdiv(nx, 256), ny), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(flow), ny, nx, Mt * Mt, s1, s2, S, rows_before,
                       rows_after);  // float64 like the reference's M_th
    HHSR_LAUNCHED();
}

// ---- frame-independent part of the noise model (robustness.py:505-528) ------------------------------
// sigma^2(p) = sum_c max(var_c(p), sigma_t(b_c(p))^2) depends only on the reference frame: computed once per
// burst (float64 sum like the reference, stored float32) instead of once per frame — the per-frame kernel
// then reads 4 raw-resolution planes (3 means + sigma^2) instead of 6.
__global__ void __launch_bounds__(256) k_rob_sigma(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                    const double* __restrict__ stdc, int ncurve,
                                                    float* __restrict__ ssq, uint32_t* __restrict__ idx, size_t n) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    double s_sq = 0.0;
    uint32_t packed = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float b = rmean[c * n + o];
        int id = 0;
        const double bb = 1000.0 * (double)b;
        if (isfinite(bb)) id = clampi((int)rint(bb), 0, ncurve - 1);  // non-finite (D6 border): see k_rob_frame
        packed |= (uint32_t)id << (10 * c);
        const double s_t = stdc[id];
        const double sp = (double)rvar[c * n + o];
        const double st2 = s_t * s_t;
        s_sq += (st2 > sp) ? st2 : sp;  // Python max(sigma_p_sq, sigma_t^2)
    }
    ssq[o] = (float)s_sq;
    if (idx) idx[o] = packed;
}

extern "C" int hhsr_rob_sigma(const float* ref_means, const float* ref_vars, int H, int W, const double* std_curve,
                              int ncurve, float* sigma_sq, uint32_t* curve_index, void* stream) {
    HHSR_ARG(ref_means && ref_vars && std_curve && sigma_sq && H > 0 && W > 0 && ncurve > 0);
    HHSR_ARG(!curve_index || ncurve <= 1024);  // three 10-bit indices per word
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(k_rob_sigma, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref_means,
                       ref_vars, std_curve, ncurve, sigma_sq, curve_index, n);
    HHSR_LAUNCHED();
}

// ---- reference-frame planes in one pass (once per burst) ----------------------------------------------------------
// hhsr_rob_upscale(means) + hhsr_rob_upscale(vars) + hhsr_rob_sigma fused: the un-warped Dodgson upsampling of the
// reference frame's guide means and variances, the noise-model sigma^2 and the packed curve indices, from guide
// tiles staged in LDS.  The upsampled variances (144 MB at 12 MP) are never written.  Same arithmetic as the
// three kernels (float64 weights, float32 buffers rounded after every tap, float64 sigma^2 sum): bit-identical.
constexpr int RP_T = 32, RP_W = RP_T / 2 + 2;  // 32 x 32 raw pixels per workgroup, 18 x 18 guide window

__global__ void __launch_bounds__(256) k_ref_planes(const float* __restrict__ gm, const float* __restrict__ gv, int lh,
                                                     int lw, const double* __restrict__ stdc, int ncurve,
                                                     float* __restrict__ rmean, float* __restrict__ ssq,
                                                     uint32_t* __restrict__ idx, int H, int W) {
    __shared__ float s_w[6][RP_W][RP_W + 1];
    const int bid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bx = (bid % gridDim.x) * RP_T, by = (bid / gridDim.x) * RP_T;
    const int wx0 = bx / 2 - 1, wy0 = by / 2 - 1;  // centre (y >> 1) - 1 of the first pixel
    const size_t gplane = (size_t)lh * lw;
    for (int p = threadIdx.x; p < 6 * RP_W * RP_W; p += 256) {
        const int c = p / (RP_W * RP_W), q = p - c * (RP_W * RP_W);
        const int i = q / RP_W, j = q - i * RP_W;
        const int gy = clampi(wy0 + i, 0, lh - 1), gx = clampi(wx0 + j, 0, lw - 1);
        s_w[c][i][j] = (c < 3 ? gm + c * gplane : gv + (c - 3) * gplane)[(size_t)gy * lw + gx];
    }
    __syncthreads();
    const size_t plane = (size_t)H * W;
    const int lx_ = threadIdx.x & 31, ly_ = threadIdx.x >> 5;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const int x = bx + lx_, y = by + ly_ + 8 * k;
        if (x >= W || y >= H) continue;
        const size_t o = (size_t)y * W + x;
        // un-warped position l = (p + 0.5) / 2 - 0.5 (robustness.py:380-383): outside for p = 0 (D6)
        const double ly = ((double)y + 0.5) / 2.0 - 0.5, lx = ((double)x + 0.5) / 2.0 - 0.5;
        float v[6];
        if (!(ly >= 0.0 && ly < (double)lh && lx >= 0.0 && lx < (double)lw)) {
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = INFINITY;
        } else if (y >= 2 && (y >> 1) + 1 <= lh - 1 && x >= 2 && (x >> 1) + 1 <= lw - 1) {
            // interior, no flow: l = p/2 -+ 0.25 by the parity of p, centre p >> 1, tap offsets (-0.75, 0.25, 1.25) or
            // (-1.25, -0.25, 0.75): the Dodgson weights are the dyadic constants 3/16, 7/8, -1/16, their products are exact
            // in float32 and sum to exactly 1, and fmaf(v, w, b) rounds the exact v w + b once — the very value the generic
            // path's float64 multiply-add followed by the float32 rounding produces.  Bit-identical, no float64.
            const int cy = y >> 1, cx = x >> 1;
            const float wyv[3] = {(y & 1) ? -0.0625f : 0.1875f, 0.875f, (y & 1) ? 0.1875f : -0.0625f};
            const float wxv[3] = {(x & 1) ? -0.0625f : 0.1875f, 0.875f, (x & 1) ? 0.1875f : -0.0625f};
            float b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float w = wyv[i] * wxv[j];
#pragma unroll
                    for (int c = 0; c < 6; ++c) b[c] = fmaf(s_w[c][cy + i - 1 - wy0][cx + j - 1 - wx0], w, b[c]);
                }
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = b[c];
        } else {
            const int cy = (int)rint(ly), cx = (int)rint(lx);
            float b[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            double wacc = 0.0;
#pragma unroll
            for (int i = -1; i <= 1; ++i) {
                const int y_ = clampi(cy + i, 0, lh - 1);
                const double wy = dodgson((double)y_ - ly);
#pragma unroll
                for (int j = -1; j <= 1; ++j) {
                    const int x_ = clampi(cx + j, 0, lw - 1);
                    const double w = wy * dodgson((double)x_ - lx);
#pragma unroll
                    for (int c = 0; c < 6; ++c)  // float32 buffer += float32 * float64, rounded after every tap
                        b[c] = (float)((double)b[c] + (double)s_w[c][cy + i - wy0][cx + j - wx0] * w);
                    wacc += w;
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) v[c] = wacc == 1.0 ? b[c] : (float)((double)b[c] / wacc);
        }
        double s_sq = 0.0;
        uint32_t packed = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rmean[c * plane + o] = v[c];
            int id = 0;
            const double bb = 1000.0 * (double)v[c];
            if (isfinite(bb)) id = clampi((int)rint(bb), 0, ncurve - 1);
            packed |= (uint32_t)id << (10 * c);
            const double s_t = stdc[id], st2 = s_t * s_t, sp = (double)v[3 + c];
            s_sq += (st2 > sp) ? st2 : sp;
        }
        ssq[o] = (float)s_sq;
        if (idx) idx[o] = packed;
    }
}

extern "C" int hhsr_ref_planes(const float* guide_means, const float* guide_vars, int lh, int lw,
                               const double* std_curve, int ncurve, float* ref_means, float* sigma_sq,
                               uint32_t* curve_index, void* stream) {
    HHSR_ARG(guide_means && guide_vars && std_curve && ref_means && sigma_sq && lh > 0 && lw > 0 && ncurve > 0);
    HHSR_ARG(!curve_index || ncurve <= 1024);
    const int H = 2 * lh, W = 2 * lw;
    hipLaunchKernelGGL(k_ref_planes, dim3(hhsr_cdiv(W, RP_T), hhsr_cdiv(H, RP_T)), dim3(256), 0, (hipStream_t)stream,
                       guide_means, guide_vars, lh, lw, std_curve, ncurve, ref_means, sigma_sq, curve_index, H, W);
    HHSR_LAUNCHED();
}

// ---- fused per-frame robustness -> R (generic tile sizes) ---------------------------------------------
__global__ void __launch_bounds__(256) k_rob_frame(const float* __restrict__ cm, int lh, int lw,
                                                    const float* __restrict__ rmean, const float* __restrict__ ssq,
                                                    const float2* __restrict__ flow, int nx, int ts,
                                                    const float* __restrict__ S, const double* __restrict__ difc,
                                                    int ncurve, double t, float* __restrict__ R, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int tix = x / ts, tiy = y / ts;
    const float2 f = flow[(size_t)tiy * nx + tix];
    float cmu[3];
    const bool inb = dodgson_sample<3>(cm, lh, lw, y, x, (double)f.x, (double)f.y, cmu);
    if (!inb) cmu[0] = cmu[1] = cmu[2] = INFINITY;
    const size_t plane = (size_t)H * W, o = (size_t)y * W + x;
    double d_sq = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float b = rmean[c * plane + o];
        const float dp = fabsf(b - cmu[c]);  // robustness.py:453-461
        // noise-model lookup at round(1000 * brightness) (robustness.py:517-520).  Non-finite brightness
        // (the +inf border of the reference map, D6) reads index 0: the reference reads out of bounds
        // there and the value never survives the NaN -> 0 clamp below.
        int id = 0;
        const double bb = 1000.0 * (double)b;
        if (isfinite(bb)) id = clampi((int)rint(bb), 0, ncurve - 1);
        const double d_t = difc[id];
        const float dp2f = dp * dp;
        const double dp2 = (double)dp2f;
        const double shrink = dp2 / (dp2 + d_t * d_t);
        d_sq += dp2 * shrink * shrink;
    }
    const float dsf = (float)d_sq, ssf = ssq[o];
    // R = clamp(S * exp(-d^2/sigma^2) - t, 0, 1), float32 up to the subtraction (robustness.py:636-639)
    const float e = expf(-dsf / ssf);
    double v = (double)(S[(size_t)tiy * nx + tix] * e) - t;
    v = v > 0.0 ? v : 0.0;  // NaN -> 0
    v = v < 1.0 ? v : 1.0;
    R[o] = (float)v;
}

// ---- the same, one 16x16 raw-pixel workgroup per flow-tile fragment, guide window in LDS -------------
// A 16-aligned 16x16 block of raw pixels lies inside one flow tile (ts is a multiple of 16), so all its
// pixels are displaced by the same flow vector and read the same <= 11x11 window of the frame's guide means.
// The window (3 channels, clamp-to-edge coordinates as in robustness.py:403-409) is staged in LDS with
// coalesced loads; the 27 taps of every pixel come from LDS: ~8 vector loads per pixel instead of 41.
constexpr int RF_T = 16;               // a 16-column group lies inside one flow tile (ts % 16 == 0)
constexpr int RF_BX = 32, RF_BY = 32;  // raw pixels per workgroup (2 x 2 sub-tiles of 16 x 16)
constexpr int RF_NK = 4;               // pixels per thread (rows ly + 8k)
constexpr int RF_WN = 11;              // guide window per sub-tile: 16/2 + 3 rows and columns

// One axis of the guide-image position of raw pixel p displaced by the tile's flow f:
//     l = (p + f + 0.5) / 2 - 0.5      (robustness.py:380-383, float64 in the reference).
// Split f = fi + ff (fi = floor(f), 0 <= ff < 1, exact in float64) and n = p + fi:  2 l = n + ff - 0.5.  The
// in-image test, round-half-even centre and the centre offset then follow from the integer n and three
// per-tile predicates on ff — the same decisions as the float64 expression without per-pixel float64.
struct RobAxis {
    int fi;       // floor(f)
    float h;      // ff / 2
    bool lt, eq;  // ff < 0.5, ff == 0.5
    bool ok;      // finite, sane flow
};
__device__ __forceinline__ RobAxis rob_axis(float f) {
    const double fd = (double)f, fl = floor(fd), ff = fd - fl;
    RobAxis a;
    a.ok = fabs(fd) < 1.0e9;  // false for NaN: such a pixel is "outside" like in the reference
    a.fi = a.ok ? (int)fl : 0;
    a.h = (float)(0.5 * ff);
    a.lt = ff < 0.5;
    a.eq = ff == 0.5;
    return a;
}
// centre c = rint(l) and r = c - l for pixel p; returns whether 0 <= l < len
__device__ __forceinline__ bool rob_centre(const RobAxis& a, int p, int len, int& c, float& r) {
    const int n = p + a.fi, m = n >> 1, odd = n & 1;
    const bool up = odd && !a.lt && (!a.eq || (m & 1));  // l = m + 0.25 + ff/2 on odd n: tie goes to the even centre
    c = m + (up ? 1 : 0);
    const float base = odd ? 0.25f + a.h : a.h - 0.25f;  // l - m
    r = (up ? 1.0f : 0.0f) - base;
    return a.ok && (n >= 1 || (n == 0 && !a.lt)) && (n < 2 * len || (n == 2 * len && a.lt));
}
// Dodgson weights of the 3 taps c-1, c, c+1 at offsets r-1, r, r+1 (|r| <= 0.5): the centre is on the inner
// branch, the neighbours on the outer one; a tap clamped to the image border coincides with the centre and
// takes the centre's weight (robustness.py:407-413).
__device__ __forceinline__ void dodgson3(float r, int c, int len, float w[3]) {
    const float inner = fmaf(-2.0f * r, r, 1.0f);
    const float am = fabsf(r - 1.0f), ap = fabsf(r + 1.0f);
    const float om = fmaf(am, am, fmaf(-2.5f, am, 1.5f)), op = fmaf(ap, ap, fmaf(-2.5f, ap, 1.5f));
    w[0] = c - 1 >= 0 ? om : inner;
    w[1] = inner;
    w[2] = c + 1 <= len - 1 ? op : inner;
}

// Arithmetic of this fused kernel (vs the reference's Numba typing):
//   * the in/out-of-image test, the window centre (round-half-even) and the clamped taps: exact integer /
//     predicate form of the reference's float64 expressions (rob_axis / rob_centre) — identical decisions;
//   * Dodgson weights and the weighted mean: float32 FMAs.  The reference evaluates the weights in
//     float64 but rounds its float32 buffer after every tap (robustness.py:414-415), so its means already
//     carry ~1e-7 relative rounding noise; float32 weights stay within that;
//   * the noise-curve indices round(1000 b_c) only depend on the reference frame: hhsr_rob_sigma takes them in
//     float64 once per burst (exact same index as the reference) and packs the three 10-bit indices in one word;
//   * the shrink d^2/(d^2 + d_t^2), the sigma^2 ratio and the exponential are float32 with v_rcp_f32 / v_exp_f32
//     (1 ulp each; relative error ~2e-7 on values that feed exp(-d^2/sigma^2)); Inf / NaN propagate like the
//     IEEE divisions they replace, so the D6 border and out-of-image pixels still end at R = 0.
// Net effect on R: <= 1e-4 absolute on the few pixels in the transition band 0 < R < 1 (tests: 1e-4).
// The kernel is VALU-bound (it was ~550 instructions per pixel, 69 of them float64, before this form).
// Workgroup = 32 x 32 raw pixels = 2 x 2 sub-tiles of 16 x 16 (each inside one flow tile, own guide window);
// a thread owns 4 pixels (rows ly, ly+8, ly+16, ly+24).  One short dependent chain (flow -> window origin ->
// guide loads -> LDS) per 1024 pixels and 20 independent plane loads in flight per thread: with one pixel per
// thread the kernel was bound by that chain's latency (3 TB/s of plane traffic at 8 workgroups per CU).
__global__ void __launch_bounds__(256) k_rob_frame_tile(const float* __restrict__ cm, int lh, int lw,
                                                         const float* __restrict__ rmean,
                                                         const float* __restrict__ ssq,
                                                         const uint32_t* __restrict__ cidx,
                                                         const float2* __restrict__ flow, int nx, int ts,
                                                         const float* __restrict__ S,
                                                         const double* __restrict__ difc, double t,
                                                         float* __restrict__ R, int H, int W) {
    __shared__ float s_g[2][2][3][RF_WN][RF_WN + 2];
    const int lx_ = threadIdx.x & (RF_BX - 1), ly_ = threadIdx.x >> 5;
    const int grp = lx_ >> 4;  // 16-column group
    const int bx = blockIdx.x * RF_BX + grp * RF_T, by = blockIdx.y * RF_BY;
    const int x = blockIdx.x * RF_BX + lx_;
    const int tix = min(bx, W - 1) / ts;
    RobAxis ay[2], ax[2];
    int wy0[2], wx0[2];
    float Sv[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {  // the two sub-tiles of this column group
        const int tiy = min(by + RF_T * v, H - 1) / ts;
        const float2 f = flow[(size_t)tiy * nx + tix];
        Sv[v] = S[(size_t)tiy * nx + tix];
        ay[v] = rob_axis(f.y);
        ax[v] = rob_axis(f.x);
        // window origin from the sub-tile's first pixel (the centre is monotone in the pixel coordinate and
        // 16 pixels advance it by at most 8, so centre-1 .. centre+1 stays inside 11 x 11 entries)
        float r_;
        rob_centre(ay[v], by + RF_T * v, lh, wy0[v], r_);
        rob_centre(ax[v], bx, lw, wx0[v], r_);
        wy0[v] = clampi(wy0[v], -4, lh + 4) - 1;
        wx0[v] = clampi(wx0[v], -4, lw + 4) - 1;
    }
    const size_t gplane = (size_t)lh * lw;
    const int tg = (lx_ & (RF_T - 1)) + RF_T * ly_;  // 0..127 within the column group
    constexpr int WSZ = 3 * RF_WN * RF_WN, NST = (WSZ + 127) / 128;
    float st[2][NST];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int p = tg + 128 * u;
            if (p < WSZ) {
                const int c = p / (RF_WN * RF_WN), q = p - c * (RF_WN * RF_WN);
                const int i = q / RF_WN, j = q - i * RF_WN;
                const int gy = clampi(wy0[v] + i, 0, lh - 1), gx = clampi(wx0[v] + j, 0, lw - 1);
                st[v][u] = cm[c * gplane + (size_t)gy * lw + gx];
            }
        }
    // the reference-frame operands do not depend on the LDS window: issue their loads before the barrier
    const size_t plane = (size_t)H * W;
    bool live[RF_NK];
    size_t o[RF_NK];
    float rb[RF_NK][3], s_sq[RF_NK];
    uint32_t ci[RF_NK];
#pragma unroll
    for (int k = 0; k < RF_NK; ++k) {
        const int y = by + ly_ + 8 * k;
        live[k] = x < W && y < H;
        o[k] = live[k] ? (size_t)y * W + x : 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) rb[k][c] = rmean[c * plane + o[k]];
        s_sq[k] = ssq[o[k]];
        ci[k] = cidx[o[k]];
    }
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int p = tg + 128 * u;
            if (p < WSZ) {
                const int c = p / (RF_WN * RF_WN), q = p - c * (RF_WN * RF_WN);
                const int i = q / RF_WN, j = q - i * RF_WN;
                s_g[grp][v][c][i][j] = st[v][u];
            }
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RF_NK; ++k) {
        if (!live[k]) continue;
        const int v = k >> 1, y = by + ly_ + 8 * k;
        float d_t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) d_t[c] = (float)difc[(ci[k] >> (10 * c)) & 1023u];
        int cy, cx;
        float ry, rx;
        const bool iny = rob_centre(ay[v], y, lh, cy, ry), inx = rob_centre(ax[v], x, lw, cx, rx);
        float cmu[3] = {INFINITY, INFINITY, INFINITY};
        if (iny && inx) {
            float wxv[3], wyv[3], b0 = 0.f, b1 = 0.f, b2 = 0.f, wacc = 0.f;
            dodgson3(rx, cx, lw, wxv);
            dodgson3(ry, cy, lh, wyv);
            const int wi0 = cy - 1 - wy0[v], wj0 = cx - 1 - wx0[v];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float w = wyv[i] * wxv[j];
                    b0 = fmaf(s_g[grp][v][0][wi0 + i][wj0 + j], w, b0);
                    b1 = fmaf(s_g[grp][v][1][wi0 + i][wj0 + j], w, b1);
                    b2 = fmaf(s_g[grp][v][2][wi0 + i][wj0 + j], w, b2);
                    wacc += w;
                }
            }
            const float iw = __builtin_amdgcn_rcpf(wacc);
            cmu[0] = b0 * iw;
            cmu[1] = b1 * iw;
            cmu[2] = b2 * iw;
        }
        float d_sq = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float dp = fabsf(rb[k][c] - cmu[c]);
            const float dp2 = dp * dp;
            const float shrink = dp2 * __builtin_amdgcn_rcpf(dp2 + d_t[c] * d_t[c]);
            d_sq += dp2 * shrink * shrink;
        }
        const float e = __builtin_amdgcn_exp2f((-d_sq * __builtin_amdgcn_rcpf(s_sq[k])) * 1.44269504088896341f);
        double r = (double)(Sv[v] * e) - t;
        r = r > 0.0 ? r : 0.0;
        r = r < 1.0 ? r : 1.0;
        R[o[k]] = (float)r;
    }
}

// The 4 horizontally adjacent pixels x0 .. x0 + 3 (x0 % 4 == 0) of one row of a sub-tile, from the sub-tile's guide
// window `win` ([3][RF_WN][RF_WN + 2], origin (wy0, wx0)): warped guide means -> colour distance -> shrink -> R.
// Pixels k and k + 2 lie one guide pixel apart with the same sub-pixel phase (l = (p + f + 0.5) / 2 - 0.5), so they
// share the Dodgson weights in x, the 9 weight products and their sum, and read a common 3 x 4 window per channel:
// the geometry / weight part is evaluated per PHASE (2 per thread) instead of per pixel and a thread reads 72 window
// values instead of 108.  Away from the left / right image border only (there the clamped taps differ per pixel and
// the per-pixel form below is used); same operations in the same order either way: bit-identical results.
__device__ __forceinline__ void rob_row4(const float (*win)[RF_WN][RF_WN + 2], const RobAxis& ay, const RobAxis& ax,
                                         int wy0, int wx0, int y, int x0, int lh, int lw, const float rbk[4][3],
                                         const float d_t2[4][3], const float iss[4], float Sv, float tf, float out[4]) {
    int cy;
    float ry, wyv[3];
    const bool iny = rob_centre(ay, y, lh, cy, ry);
    dodgson3(ry, cy, lh, wyv);
    const int wi0 = cy - 1 - wy0;
    float cmu[4][3];
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {  // pixels ph and ph + 2
        int cx;
        float rx;
        const bool inx = rob_centre(ax, x0 + ph, lw, cx, rx);
#pragma unroll
        for (int c = 0; c < 3; ++c) cmu[ph][c] = cmu[ph + 2][c] = INFINITY;
        // pixel ph + 2: l + 1, centre cx + 1, same offset rx; inside the image when cx + 2 <= lw - 1 (l + 1 <= cx + 1.5)
        if (iny && inx && cx >= 1 && cx + 2 <= lw - 1) {  // no clamped tap in x for either pixel
            float wxv[3], w[3][3], wacc = 0.f;
            dodgson3(rx, cx, lw, wxv);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    w[i][j] = wyv[i] * wxv[j];
                    wacc += w[i][j];
                }
            const float iw = __builtin_amdgcn_rcpf(wacc);
            const int wj0 = cx - 1 - wx0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float g4[3][4];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) g4[i][j] = win[c][wi0 + i][wj0 + j];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float b = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) b = fmaf(g4[i][j + q], w[i][j], b);
                    cmu[ph + 2 * q][c] = b * iw;
                }
            }
        } else {
            int cx2;
            float rx2;
            const bool inx2 = rob_centre(ax, x0 + ph + 2, lw, cx2, rx2);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int cxq = q ? cx2 : cx;
                const float rxq = q ? rx2 : rx;
                if (iny && (q ? inx2 : inx)) {
                    float wxv[3], b0 = 0.f, b1 = 0.f, b2 = 0.f, wacc = 0.f;
                    dodgson3(rxq, cxq, lw, wxv);
                    const int wj0 = cxq - 1 - wx0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float w = wyv[i] * wxv[j];
                            b0 = fmaf(win[0][wi0 + i][wj0 + j], w, b0);
                            b1 = fmaf(win[1][wi0 + i][wj0 + j], w, b1);
                            b2 = fmaf(win[2][wi0 + i][wj0 + j], w, b2);
                            wacc += w;
                        }
                    }
                    const float iw = __builtin_amdgcn_rcpf(wacc);
                    cmu[ph + 2 * q][0] = b0 * iw;
                    cmu[ph + 2 * q][1] = b1 * iw;
                    cmu[ph + 2 * q][2] = b2 * iw;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float d_sq = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float dp = fabsf(rbk[k][c] - cmu[k][c]);
            const float dp2 = dp * dp;
            const float shrink = dp2 * __builtin_amdgcn_rcpf(dp2 + d_t2[k][c]);
            d_sq += dp2 * shrink * shrink;
        }
        const float e = __builtin_amdgcn_exp2f((-d_sq * iss[k]) * 1.44269504088896341f);
        // R = clamp(S e - t, 0, 1) (robustness.py:636-639; the reference subtracts in float64 and rounds: <= 1 ulp of R
        // apart); v_med3_f32 returns min(0, 1) for a NaN operand = the reference's NaN -> 0
        out[k] = __builtin_amdgcn_fmed3f(Sv * e - tf, 0.f, 1.f);
    }
}

// Same kernel with a ROW mapping: a thread owns 4 horizontally adjacent pixels of one row (one flow tile, one
// sub-tile window).  The five reference-frame planes and R move as 16-byte vectors (1 KB per wave instruction
// instead of 256 B: the dword version streamed at 3.5 TB/s), and the row part of the geometry — in-image test,
// centre, Dodgson weights in y — is evaluated once per 4 pixels.  Needs W % 4 == 0 and 16-byte aligned planes.
__global__ void __launch_bounds__(256) k_rob_frame_row4(const float* __restrict__ cm, int lh, int lw,
                                                         const float* __restrict__ rmean,
                                                         const float* __restrict__ ssq,
                                                         const uint32_t* __restrict__ cidx,
                                                         const float2* __restrict__ flow, int nx, int ts,
                                                         const float* __restrict__ S,
                                                         const double* __restrict__ difc, double t,
                                                         float* __restrict__ R, int H, int W) {
    __shared__ float s_g[2][2][3][RF_WN][RF_WN + 2];
    const int lx4 = threadIdx.x & 7, ly_ = threadIdx.x >> 3;  // 8 threads x 4 pixels per row, 32 rows
    const int grp = lx4 >> 2, v = ly_ >> 4;                   // the thread's 16 x 16 sub-tile
    const int bid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);  // bands of tile rows per XCD
    const int bxi = bid % gridDim.x, byi = bid / gridDim.x;
    const int sx0 = bxi * RF_BX + grp * RF_T, sy0 = byi * RF_BY + v * RF_T;
    const int x0 = bxi * RF_BX + 4 * lx4, y = byi * RF_BY + ly_;
    const int tix = min(sx0, W - 1) / ts, tiy = min(sy0, H - 1) / ts;
    const float2 f = flow[(size_t)tiy * nx + tix];
    const float Sv = S[(size_t)tiy * nx + tix];
    const RobAxis ay = rob_axis(f.y), ax = rob_axis(f.x);
    int wy0, wx0;
    {
        float r_;
        rob_centre(ay, sy0, lh, wy0, r_);
        rob_centre(ax, sx0, lw, wx0, r_);
        wy0 = clampi(wy0, -4, lh + 4) - 1;
        wx0 = clampi(wx0, -4, lw + 4) - 1;
    }
    const size_t gplane = (size_t)lh * lw;
    const int tg = (ly_ & (RF_T - 1)) * 4 + (lx4 & 3);  // 0..63 within the sub-tile
    constexpr int WSZ = 3 * RF_WN * RF_WN, NST = (WSZ + 63) / 64;
    float st[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int p = tg + 64 * u;
        if (p < WSZ) {
            const int c = p / (RF_WN * RF_WN), q = p - c * (RF_WN * RF_WN);
            const int i = q / RF_WN, j = q - i * RF_WN;
            const int gy = clampi(wy0 + i, 0, lh - 1), gx = clampi(wx0 + j, 0, lw - 1);
            st[u] = cm[c * gplane + (size_t)gy * lw + gx];
        }
    }
    const size_t plane = (size_t)H * W;
    const bool live = x0 < W && y < H;
    const size_t o = live ? (size_t)y * W + x0 : 0;
    float4 rb4[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rb4[c] = *reinterpret_cast<const float4*>(rmean + c * plane + o);
    const float4 ss4 = *reinterpret_cast<const float4*>(ssq + o);
    const uint4 ci4 = *reinterpret_cast<const uint4*>(cidx + o);
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int p = tg + 64 * u;
        if (p < WSZ) {
            const int c = p / (RF_WN * RF_WN), q = p - c * (RF_WN * RF_WN);
            const int i = q / RF_WN, j = q - i * RF_WN;
            s_g[grp][v][c][i][j] = st[u];
        }
    }
    __syncthreads();
    if (!live) return;
    const float rbk[4][3] = {{rb4[0].x, rb4[1].x, rb4[2].x}, {rb4[0].y, rb4[1].y, rb4[2].y},
                             {rb4[0].z, rb4[1].z, rb4[2].z}, {rb4[0].w, rb4[1].w, rb4[2].w}};
    const float ssk[4] = {ss4.x, ss4.y, ss4.z, ss4.w};
    const uint32_t cik[4] = {ci4.x, ci4.y, ci4.z, ci4.w};
    float d_t2[4][3], iss[4], out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = (float)difc[(cik[k] >> (10 * c)) & 1023u];
            d_t2[k][c] = d * d;
        }
        iss[k] = __builtin_amdgcn_rcpf(ssk[k]);
    }
    rob_row4(s_g[grp][v], ay, ax, wy0, wx0, y, x0, lh, lw, rbk, d_t2, iss, Sv, (float)t, out);
    *reinterpret_cast<float4*>(R + o) = make_float4(out[0], out[1], out[2], out[3]);
}

// ---- several frames per launch ------------------------------------------------------------------------------------
// The reference-frame operands — three upsampled mean planes, sigma^2, the packed curve indices: 20 of the 27 bytes a
// frame moves per pixel — do not depend on the compared frame, and k_rob_frame_row4 streams them again for every frame of
// the burst (326 MB per launch at 12 MP, 4.3 TB/s: HBM-bound).  This variant keeps them (and the three d_t lookups
// per pixel) in registers and loops over up to ROB_GROUP frames of the same burst: per frame only the guide-means
// window (3 B / pixel) comes in and R (4 B / pixel) goes out.  Same arithmetic per frame: bit-identical to one launch
// of k_rob_frame_row4 per frame.
constexpr int ROB_GROUP = 4;
struct RobGroup {
    const float* cm[ROB_GROUP];
    const float2* flow[ROB_GROUP];
    const float* S[ROB_GROUP];
    float* R[ROB_GROUP];
    int n;
};

// (no occupancy bound: held to 64 VGPRs — 8 waves per SIMD — the kernel spills and takes 505 instead of 222 us per launch;
// held to 80: 423 us.  The compiler's own choice is 56 VGPRs.)
// Measured without gain (round 5, after round 4's counters showed 54 % of the wave time parked with the VALU 78 % busy): a
// wave-private mapping — wave = one 16 x 16 sub-tile, lane = (row, 4-pixel group), the guide window double-buffered in LDS,
// the two workgroup barriers per frame replaced by wave-level fences: 218.7 against 218.5 us per 4-frame launch at 12 MP
// (60 against 56 VGPRs, 14.2 against 7.4 KB of LDS; bit-identical).  The waves do not wait for each other at those
// barriers; what they wait for is their own dependent chains (27 taps x 4 pixels of FMAs behind 72 LDS reads, 12 v_rcp_f32,
// 4 v_exp_f32 per thread and frame) — the kernel is at ~6 cycles per VALU instruction like the other tap kernels.
__global__ void __launch_bounds__(256) k_rob_frames_row4(RobGroup gq, int lh, int lw, const float* __restrict__ rmean,
                                                          const float* __restrict__ ssq,
                                                          const uint32_t* __restrict__ cidx, int ny, int nx, int ts,
                                                          const double* __restrict__ difc, double t, int H, int W,
                                                          double Mt2, float s1, float s2, int rows_lo, int rows_hi) {
    __shared__ float s_g[2][2][3][RF_WN][RF_WN + 2];
    __shared__ float4 s_tab[ROB_GROUP][2][2][2];  // per (frame, sub-tile): flow split, window origin, S — see below
    const int lx4 = threadIdx.x & 7, ly_ = threadIdx.x >> 3;  // 8 threads x 4 pixels per row, 32 rows
    const int grp = lx4 >> 2, v = ly_ >> 4;                   // the thread's 16 x 16 sub-tile
    const int bid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);  // bands of tile rows per XCD
    const int bxi = bid % gridDim.x, byi = bid / gridDim.x;
    const int x0 = bxi * RF_BX + 4 * lx4, y = byi * RF_BY + ly_;
    const size_t gplane = (size_t)lh * lw, plane = (size_t)H * W;
    const int tg = (ly_ & (RF_T - 1)) * 4 + (lx4 & 3);  // 0..63 within the sub-tile
    constexpr int WSZ = 3 * RF_WN * RF_WN, NST = (WSZ + 63) / 64;
    const bool live = x0 < W && y < H;
    const size_t o = live ? (size_t)y * W + x0 : 0;
    // per-burst operands of the thread's 4 pixels, once
    float4 rb4[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rb4[c] = *reinterpret_cast<const float4*>(rmean + c * plane + o);
    const float4 ss4 = *reinterpret_cast<const float4*>(ssq + o);
    const uint4 ci4 = *reinterpret_cast<const uint4*>(cidx + o);
    const float rbk[4][3] = {{rb4[0].x, rb4[1].x, rb4[2].x}, {rb4[0].y, rb4[1].y, rb4[2].y},
                             {rb4[0].z, rb4[1].z, rb4[2].z}, {rb4[0].w, rb4[1].w, rb4[2].w}};
    const float ssk[4] = {ss4.x, ss4.y, ss4.z, ss4.w};
    const uint32_t cik[4] = {ci4.x, ci4.y, ci4.z, ci4.w};
    float d_t2[4][3], iss[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = (float)difc[(cik[k] >> (10 * c)) & 1023u];
            d_t2[k][c] = d * d;
        }
        iss[k] = __builtin_amdgcn_rcpf(ssk[k]);
    }
    // The flow split (float64 floor / compares), the window origin and S only depend on (frame, sub-tile): 16 lanes
    // evaluate them once per workgroup instead of every thread for every frame (~150 cycles of half-rate work per thread
    // and frame, 11 % of the kernel); the frame loop reads them back as two 16-byte LDS words.
    if (threadIdx.x < gq.n * 4) {
        const int fr = threadIdx.x >> 2, vv = (threadIdx.x >> 1) & 1, gg = threadIdx.x & 1;
        const int tsx0 = bxi * RF_BX + gg * RF_T, tsy0 = byi * RF_BY + vv * RF_T;
        const size_t tl = (size_t)(min(tsy0, H - 1) / ts) * nx + min(tsx0, W - 1) / ts;
        const float2 f = gq.flow[fr][tl];
        const RobAxis ay = rob_axis(f.y), ax = rob_axis(f.x);
        int wy0, wx0;
        float r_;
        rob_centre(ay, tsy0, lh, wy0, r_);
        rob_centre(ax, tsx0, lw, wx0, r_);
        wy0 = clampi(wy0, -4, lh + 4) - 1;
        wx0 = clampi(wx0, -4, lw + 4) - 1;
        const int flags = (ay.lt ? 1 : 0) | (ay.eq ? 2 : 0) | (ay.ok ? 4 : 0) | (ax.lt ? 8 : 0) | (ax.eq ? 16 : 0) | (ax.ok ? 32 : 0);
        s_tab[fr][vv][gg][0] = make_float4(__int_as_float(ay.fi), ay.h, __int_as_float(wy0), __int_as_float(flags));
        float Sv;
        if (gq.S[fr]) {
            Sv = gq.S[fr][tl];
        } else {  // the flow-irregularity weight of the tile (k_rob_s, robustness.py:570-612), evaluated here
            const int tiy = (int)(tl / nx), tix = (int)(tl - (size_t)tiy * nx);
            float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
            for (int i = -1; i <= 1; ++i)
                for (int j = -1; j <= 1; ++j) {
                    const int yy = tiy + i, xx = tix + j;
                    if (yy >= -rows_lo && yy < ny + rows_hi && xx >= 0 && xx < nx) {  // (see k_rob_s)
                        const float2 q = gq.flow[fr][(ptrdiff_t)yy * nx + xx];
                        mxx = fmaxf(mxx, q.x); mxy = fmaxf(mxy, q.y);
                        mnx = fminf(mnx, q.x); mny = fminf(mny, q.y);
                    }
                }
            const float d0 = mxx - mnx, d1 = mxy - mny;
            const float m = d0 * d0 + d1 * d1;
            Sv = ((double)m > Mt2) ? s1 : s2;
        }
        s_tab[fr][vv][gg][1] = make_float4(__int_as_float(ax.fi), ax.h, __int_as_float(wx0), Sv);
    }
    __syncthreads();
    // the guide-means window of frame fr + 1 is fetched into registers while frame fr is evaluated (round 4: the loads
    // used to be issued at the top of their own iteration and waited for right away — 56 % of the wave time was parked;
    // 228 -> 222 us per 4-frame launch at 12 MP)
    float st[NST];
    // slot p = tg + 64 u of the window is (channel c, row i, column j) for every frame: decomposed once; the address is
    // 32-bit (the three guide planes are far below 4 GB) on the frame's plane pointer in SGPRs (round 6: the 64-bit form
    // and the per-frame divisions were ~100 of the kernel's ~540 VALU instructions per thread and frame)
    int fci[NST];  // c * plane + (i << 16 | j) does not fit one register: two small fields + the plane offset
    unsigned fco[NST];
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int p = tg + 64 * u;
        const int c = p / (RF_WN * RF_WN), q = p - c * (RF_WN * RF_WN);
        const int i = q / RF_WN, j = q - i * RF_WN;
        fci[u] = (i << 8) | j;
        fco[u] = (unsigned)c * (unsigned)gplane;
    }
    auto fetch = [&](int fr) {
        const float* __restrict__ cm = gq.cm[fr];
        const int wy0 = __float_as_int(s_tab[fr][v][grp][0].z), wx0 = __float_as_int(s_tab[fr][v][grp][1].z);
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int p = tg + 64 * u;
            if (p < WSZ) {
                const int gy = clampi(wy0 + (fci[u] >> 8), 0, lh - 1), gx = clampi(wx0 + (fci[u] & 255), 0, lw - 1);
                const unsigned off = (fco[u] + (unsigned)(__mul24(gy, lw) + gx)) * 4u;  // (gy, lw < 2^24)
                st[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(cm) + off);
            }
        }
    };
    if (gq.n > 0) fetch(0);
    for (int fr = 0; fr < gq.n; ++fr) {
        const float4 t0 = s_tab[fr][v][grp][0], t1 = s_tab[fr][v][grp][1];
        const int flags = __float_as_int(t0.w);
        RobAxis ay, ax;
        ay.fi = __float_as_int(t0.x); ay.h = t0.y; ay.lt = flags & 1; ay.eq = flags & 2; ay.ok = flags & 4;
        ax.fi = __float_as_int(t1.x); ax.h = t1.y; ax.lt = flags & 8; ax.eq = flags & 16; ax.ok = flags & 32;
        const int wy0 = __float_as_int(t0.z), wx0 = __float_as_int(t1.z);
        const float Sv = t1.w;
        if (fr) __syncthreads();  // the previous frame's taps are done with the window
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int p = tg + 64 * u;
            if (p < WSZ) {
                const int c = p / (RF_WN * RF_WN), q = p - c * (RF_WN * RF_WN);
                const int i = q / RF_WN, j = q - i * RF_WN;
                s_g[grp][v][c][i][j] = st[u];
            }
        }
        __syncthreads();
        if (fr + 1 < gq.n) fetch(fr + 1);
        if (!live) continue;
        float out[4];
        rob_row4(s_g[grp][v], ay, ax, wy0, wx0, y, x0, lh, lw, rbk, d_t2, iss, Sv, (float)t, out);
        *reinterpret_cast<float4*>(gq.R[fr] + o) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

extern "C" int hhsr_rob_frame(const float* comp_means, int lh, int lw, const float* ref_means,
                              const float* ref_sigma_sq, const uint32_t* ref_curve_index, const float* flow, int ny,
                              int nx, int ts, const float* S, const double* diff_curve, int ncurve, double t, float* R,
                              void* stream) {
    HHSR_ARG(comp_means && ref_means && ref_sigma_sq && flow && S && diff_curve && R);
    HHSR_ARG(lh > 0 && lw > 0 && ts > 0 && ncurve > 0);
    const int H = 2 * lh, W = 2 * lw;
    HHSR_ARG(ny * ts >= H && nx * ts >= W);
    static const bool no_row4 = getenv("HHSR_ROB_NO_ROW4") != nullptr;  // read once
    const bool vec4 = W % 4 == 0 && (((uintptr_t)ref_means | (uintptr_t)ref_sigma_sq | (uintptr_t)ref_curve_index |
                                      (uintptr_t)R) & 15) == 0 && !no_row4;
    if (ts % RF_T == 0 && ref_curve_index && ncurve <= 1024 && vec4)
        hipLaunchKernelGGL(k_rob_frame_row4, dim3(hhsr_cdiv(W, RF_BX), hhsr_cdiv(H, RF_BY)), dim3(256), 0,
                           (hipStream_t)stream, comp_means, lh, lw, ref_means, ref_sigma_sq, ref_curve_index,
                           reinterpret_cast<const float2*>(flow), nx, ts, S, diff_curve, t, R, H, W);
    else if (ts % RF_T == 0 && ref_curve_index && ncurve <= 1024)
        hipLaunchKernelGGL(k_rob_frame_tile, dim3(hhsr_cdiv(W, RF_BX), hhsr_cdiv(H, RF_BY)), dim3(256), 0,
                           (hipStream_t)stream, comp_means, lh, lw, ref_means, ref_sigma_sq, ref_curve_index,
                           reinterpret_cast<const float2*>(flow), nx, ts, S, diff_curve, t, R, H, W);
    else
        hipLaunchKernelGGL(k_rob_frame, dim3(hhsr_cdiv(W, 64), hhsr_cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream,
                           comp_means, lh, lw, ref_means, ref_sigma_sq, reinterpret_cast<const float2*>(flow), nx, ts,
                           S, diff_curve, ncurve, t, R, H, W);
    HHSR_LAUNCHED();
}

extern "C" int hhsr_rob_frames(const float* const* comp_means, int n_frames, int lh, int lw, const float* ref_means,
                               const float* ref_sigma_sq, const uint32_t* ref_curve_index, const float* const* flows,
                               int ny, int nx, int ts, const float* const* S, double Mt, float s1, float s2,
                               const double* diff_curve, int ncurve, double t, float* const* R, int flow_rows_before,
                               int flow_rows_after, void* stream) {
    HHSR_ARG(comp_means && flows && R && n_frames >= 0 && flow_rows_before >= 0 && flow_rows_after >= 0);
    for (int n = 0; n < n_frames; ++n) HHSR_ARG(comp_means[n] && flows[n] && (!S || S[n]) && R[n]);
    HHSR_ARG(ref_means && ref_sigma_sq && diff_curve && lh > 0 && lw > 0 && ts > 0 && ncurve > 0);
    const int H = 2 * lh, W = 2 * lw;
    HHSR_ARG(ny * ts >= H && nx * ts >= W);
    static const bool no_group = getenv("HHSR_ROB_NO_GROUP") != nullptr;  // A/B switch, read once
    bool vec4 = W % 4 == 0 && (((uintptr_t)ref_means | (uintptr_t)ref_sigma_sq | (uintptr_t)ref_curve_index) & 15) == 0;
    for (int n = 0; n < n_frames; ++n) vec4 = vec4 && ((uintptr_t)R[n] & 15) == 0;
    if (no_group || !(ts % RF_T == 0 && ref_curve_index && ncurve <= 1024 && vec4)) {
        if (!S) {
            hhsr_set_error("hhsr_rob_frames: S = NULL (weights evaluated inside the kernel) needs the grouped kernel: "
                           "ts %% 16 == 0, W %% 4 == 0, packed curve indices, 16-byte aligned planes");
            return -3;
        }
        for (int n = 0; n < n_frames; ++n) {
            const int rc = hhsr_rob_frame(comp_means[n], lh, lw, ref_means, ref_sigma_sq, ref_curve_index, flows[n], ny, nx,
                                          ts, S[n], diff_curve, ncurve, t, R[n], stream);
            if (rc) return rc;
        }
        return 0;
    }
    for (int n0 = 0; n0 < n_frames; n0 += ROB_GROUP) {
        RobGroup g;
        g.n = n_frames - n0 < ROB_GROUP ? n_frames - n0 : ROB_GROUP;
        for (int k = 0; k < ROB_GROUP; ++k) {
            const int n = n0 + (k < g.n ? k : 0);
            g.cm[k] = comp_means[n];
            g.flow[k] = reinterpret_cast<const float2*>(flows[n]);
            g.S[k] = S ? S[n] : nullptr;
            g.R[k] = R[n];
        }
        hipLaunchKernelGGL(k_rob_frames_row4, dim3(hhsr_cdiv(W, RF_BX), hhsr_cdiv(H, RF_BY)), dim3(256), 0,
                           (hipStream_t)stream, g, lh, lw, ref_means, ref_sigma_sq, ref_curve_index, ny, nx, ts, diff_curve, t,
                           H, W, Mt * Mt, s1, s2, flow_rows_before, flow_rows_after);
    }
    HHSR_LAUNCHED();
}

// ---- 5x5 local minimum (robustness.py:670-686) ------------------------------------------------------
constexpr int LM_TX = 64, LM_TY = 16;

__global__ void __launch_bounds__(256) k_local_min5(const float* __restrict__ R, int H, int W,
                                                     float* __restrict__ r, float* __restrict__ acc) {
    __shared__ float s[LM_TY + 4][LM_TX + 4 + 1];
    __shared__ float s_row[LM_TY + 4][LM_TX + 1];
    const int x0 = blockIdx.x * LM_TX, y0 = blockIdx.y * LM_TY;
    for (int p = threadIdx.x; p < (LM_TY + 4) * (LM_TX + 4); p += 256) {
        const int i = p / (LM_TX + 4), j = p - i * (LM_TX + 4);
        s[i][j] = R[(size_t)clampi(y0 + i - 2, 0, H - 1) * W + clampi(x0 + j - 2, 0, W - 1)];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < (LM_TY + 4) * LM_TX; p += 256) {
        const int i = p / LM_TX, j = p - i * LM_TX;
        float m = s[i][j];
#pragma unroll
        for (int k = 1; k < 5; ++k) m = fminf(m, s[i][j + k]);
        s_row[i][j] = m;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < LM_TY * LM_TX; p += 256) {
        const int i = p / LM_TX, j = p - i * LM_TX;
        const int y = y0 + i, x = x0 + j;
        if (y < H && x < W) {
            float m = s_row[i][j];
#pragma unroll
            for (int k = 1; k < 5; ++k) m = fminf(m, s_row[i + k][j]);
            r[(size_t)y * W + x] = m;
            if (acc) acc[(size_t)y * W + x] += m;  // accumulated robustness (super_resolution.py:158-159)
        }
    }
}

extern "C" int hhsr_local_min5(const float* R, int H, int W, float* r, float* acc_r, void* stream) {
    HHSR_ARG(R && r && H > 0 && W > 0 && R != r);
    hipLaunchKernelGGL(k_local_min5, dim3(hhsr_cdiv(W, LM_TX), hhsr_cdiv(H, LM_TY)), dim3(256), 0,
                       (hipStream_t)stream, R, H, W, r, acc_r);
    HHSR_LAUNCHED();
}

// ---- accumulated robustness as the reference keeps it where something decides on it (super_resolution.py:116-117,
// 158-159; utils.py:93-120; merge.py:223-228): the float64 sum of the frames' maps, in frame order.  One pass over the n maps
// instead of n read-modify-write passes of a float64 torch tensor between the frames (round 5: 4.6 GB of extra traffic per
// 12 MP x 20 burst and five elementwise passes for the decision map).  Outputs, each optional: the float64 sum (kept for
// bursts longer than one call: `load`), the sum rounded to float32 (the map the API reports), and the float32 DECISION map a
// with  a <= mfc  <=>  sum <= mfc  and  a < mfc  <=>  sum < mfc  for the kernels that compare (double) a with
// max_frame_count (robustness.RobustnessSum.decisions_of: exact for every threshold float32 can hold).
struct RobSumFrames {
    const float* r[HHSR_MAX_FRAMES];
    int n;
};

__global__ void __launch_bounds__(256) k_rob_sum(RobSumFrames fr, size_t count, int load, double mfc, float m32, float below,
                                                  float above, double* __restrict__ sum64, float* __restrict__ mask32,
                                                  float* __restrict__ dec32) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double s = load ? sum64[i] : 0.0;
    for (int n = 0; n < fr.n; ++n) s += (double)fr.r[n][i];  // float32 -> float64: exact; frame order like the reference
    if (sum64) sum64[i] = s;
    float a = (float)s;
    if (mask32) mask32[i] = a;
    if (dec32) {
        if (s < mfc && a >= m32) a = below;
        if (s > mfc && a <= m32) a = above;
        dec32[i] = a;
    }
}

// The same over UN-filtered maps R: the 5 x 5 clamp-border minimum of every frame (robustness.py:641-686, k_local_min5's
// two LDS passes) is taken here, on the way into the sum — for callers whose merge applies the minimum itself
// (HHSR_MERGE_LOCAL_MIN) and that never materialise the filtered maps: one pass over the n maps instead of n launches of
// k_local_min5 (a read and a write of every map) plus k_rob_sum's read.  The next frame's tile is fetched into registers
// while the current one is reduced.
__global__ void __launch_bounds__(256) k_rob_sum_min5(RobSumFrames fr, int H, int W, int load, double mfc, float m32,
                                                       float below, float above, double* __restrict__ sum64,
                                                       float* __restrict__ mask32, float* __restrict__ dec32) {
    __shared__ float s[LM_TY + 4][LM_TX + 4 + 1];
    __shared__ float s_row[LM_TY + 4][LM_TX + 1];
    constexpr int NIN = ((LM_TY + 4) * (LM_TX + 4) + 255) / 256, NROW = (LM_TY + 4) * LM_TX / 256, NOUT = LM_TY * LM_TX / 256;
    const int x0 = blockIdx.x * LM_TX, y0 = blockIdx.y * LM_TY, tid = threadIdx.x;
    unsigned off[NIN];  // clamped source offsets of the thread's tile slots: the same for every frame
#pragma unroll
    for (int u = 0; u < NIN; ++u) {
        const int p = tid + 256 * u, i = p / (LM_TX + 4), j = p - i * (LM_TX + 4);
        off[u] = (unsigned)(clampi(y0 + i - 2, 0, H - 1) * W + clampi(x0 + j - 2, 0, W - 1));
    }
    double acc[NOUT];
#pragma unroll
    for (int u = 0; u < NOUT; ++u) {
        const int p = tid + 256 * u, i = p / LM_TX, j = p - i * LM_TX;
        const bool live = y0 + i < H && x0 + j < W;
        acc[u] = (load && live) ? sum64[(size_t)(y0 + i) * W + x0 + j] : 0.0;
    }
    float pre[NIN];
    auto fetch = [&](int n) {
#pragma unroll
        for (int u = 0; u < NIN; ++u)
            if (tid + 256 * u < (LM_TY + 4) * (LM_TX + 4)) pre[u] = fr.r[n][off[u]];
    };
    if (fr.n > 0) fetch(0);
    for (int n = 0; n < fr.n; ++n) {
#pragma unroll
        for (int u = 0; u < NIN; ++u) {
            const int p = tid + 256 * u, i = p / (LM_TX + 4), j = p - i * (LM_TX + 4);
            if (p < (LM_TY + 4) * (LM_TX + 4)) s[i][j] = pre[u];
        }
        __syncthreads();  // (also: the previous frame's column pass is done with s_row)
        if (n + 1 < fr.n) fetch(n + 1);
#pragma unroll
        for (int u = 0; u < NROW; ++u) {
            const int p = tid + 256 * u, i = p / LM_TX, j = p - i * LM_TX;
            float m = s[i][j];
#pragma unroll
            for (int k = 1; k < 5; ++k) m = fminf(m, s[i][j + k]);
            s_row[i][j] = m;
        }
        __syncthreads();  // (also: the row pass is done with s, which the next frame overwrites)
#pragma unroll
        for (int u = 0; u < NOUT; ++u) {
            const int p = tid + 256 * u, i = p / LM_TX, j = p - i * LM_TX;
            float m = s_row[i][j];
#pragma unroll
            for (int k = 1; k < 5; ++k) m = fminf(m, s_row[i + k][j]);
            acc[u] += (double)m;  // float32 -> float64: exact; frame order like the reference
        }
    }
#pragma unroll
    for (int u = 0; u < NOUT; ++u) {
        const int p = tid + 256 * u, i = p / LM_TX, j = p - i * LM_TX;
        if (y0 + i >= H || x0 + j >= W) continue;
        const size_t o = (size_t)(y0 + i) * W + x0 + j;
        const double sv = acc[u];
        if (sum64) sum64[o] = sv;
        float a = (float)sv;
        if (mask32) mask32[o] = a;
        if (dec32) {
            if (sv < mfc && a >= m32) a = below;
            if (sv > mfc && a <= m32) a = above;
            dec32[o] = a;
        }
    }
}
static_assert((LM_TY + 4) * LM_TX % 256 == 0 && LM_TY * LM_TX % 256 == 0, "k_rob_sum_min5: whole passes per thread");

extern "C" int hhsr_rob_sum(const float* const* rs, int n_frames, int H, int W, int flags, double max_frame_count,
                            double* sum64, float* mask32, float* decisions32, void* stream) {
    const int load = flags & HHSR_ROB_SUM_LOAD;
    HHSR_ARG(n_frames >= 0 && n_frames <= HHSR_MAX_FRAMES && H > 0 && W > 0);
    HHSR_ARG(!(flags & ~(HHSR_ROB_SUM_LOAD | HHSR_ROB_SUM_MIN5)));
    HHSR_ARG(n_frames == 0 || rs != nullptr);
    HHSR_ARG(!load || sum64 != nullptr);
    HHSR_ARG(sum64 || mask32 || decisions32);
    RobSumFrames fr;
    fr.n = n_frames;
    for (int k = 0; k < HHSR_MAX_FRAMES; ++k) {
        fr.r[k] = k < n_frames ? rs[k] : nullptr;
        HHSR_ARG(k >= n_frames || rs[k] != nullptr);
    }
    const float m32 = (float)max_frame_count;
    // largest float32 below / smallest float32 above the threshold (the threshold itself where float32 cannot hold it)
    const float below = (double)m32 < max_frame_count ? m32 : nextafterf(m32, -INFINITY);
    const float above = (double)m32 > max_frame_count ? m32 : nextafterf(m32, INFINITY);
    const size_t count = (size_t)H * W;
    if ((flags & HHSR_ROB_SUM_MIN5) && n_frames > 0) {
        HHSR_ARG(count < ((size_t)1 << 32));  // (32-bit element offsets)
        hipLaunchKernelGGL(k_rob_sum_min5, dim3(hhsr_cdiv(W, LM_TX), hhsr_cdiv(H, LM_TY)), dim3(256), 0, (hipStream_t)stream,
                           fr, H, W, load, max_frame_count, m32, below, above, sum64, mask32, decisions32);
        HHSR_LAUNCHED();
    }
    hipLaunchKernelGGL(k_rob_sum, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fr, count, load,
                       max_frame_count, m32, below, above, sum64, mask32, decisions32);
    HHSR_LAUNCHED();
}

// ---- monochrome sensors (`mode: grey`) ---------------------------------------------------------------------------
// The frame itself is the one-channel guide image (robustness.py:62-66, 145-148); its 3x3 statistics come from
// hhsr_mono_frame_stats.  The statistics map keeps its size in this mode (robustness.py:337-343) while the upscale
// kernel still divides the position by its hard-coded s = 2 (robustness.py:358): every consumer sees the top-left
// quadrant of the map stretched over the frame, the reference frame and the warped frames alike.  Deterministic, so
// reproduced: the kernels below are the reference's float64 arithmetic on that geometry, one thread per pixel, the
// Dodgson taps from L2 (a monochrome burst does not have the Bayer path's fused / LDS-staged variants).
__global__ void __launch_bounds__(256) k_mono_rob_upscale(const float* __restrict__ LR, int H, int W,
                                                           const float2* __restrict__ flow, int nx, int ts,
                                                           float* __restrict__ HR) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    double fx = 0.0, fy = 0.0;
    if (flow) {
        const float2 f = flow[(size_t)(y / ts) * nx + x / ts];
        fx = (double)f.x;
        fy = (double)f.y;
    }
    float v[1];
    if (!dodgson_sample<1>(LR, H, W, y, x, fx, fy, v)) v[0] = INFINITY;
    HR[(size_t)y * W + x] = v[0];
}

extern "C" int hhsr_mono_rob_upscale(const float* stats, int H, int W, const float* flow, int ny, int nx, int ts,
                                     float* out, void* stream) {
    HHSR_ARG(stats && out && H > 0 && W > 0);
    if (flow) HHSR_ARG(ts > 0 && (int64_t)ny * ts >= H && (int64_t)nx * ts >= W);
    hipLaunchKernelGGL(k_mono_rob_upscale, dim3(hhsr_cdiv(W, 64), hhsr_cdiv(H, 4)), dim3(256), 0, (hipStream_t)stream,
                       stats, H, W, reinterpret_cast<const float2*>(flow), nx, ts > 0 ? ts : 1, out);
    HHSR_LAUNCHED();
}

__global__ void __launch_bounds__(256) k_mono_rob_sigma(const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                         const double* __restrict__ stdc, int ncurve,
                                                         float* __restrict__ ssq, size_t n) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    int id = 0;
    const double bb = 1000.0 * (double)rmean[o];
    if (isfinite(bb)) id = clampi((int)rint(bb), 0, ncurve - 1);  // non-finite (D6 border): see k_rob_frame
    const double s_t = stdc[id], st2 = s_t * s_t, sp = (double)rvar[o];
    ssq[o] = (float)(0.0 + ((st2 > sp) ? st2 : sp));  // Python max(sigma_p_sq, sigma_t^2)
}

extern "C" int hhsr_mono_rob_sigma(const float* ref_means, const float* ref_vars, int H, int W,
                                   const double* std_curve, int ncurve, float* sigma_sq, void* stream) {
    HHSR_ARG(ref_means && ref_vars && std_curve && sigma_sq && H > 0 && W > 0 && ncurve > 0);
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(k_mono_rob_sigma, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ref_means, ref_vars, std_curve, ncurve, sigma_sq, n);
    HHSR_LAUNCHED();
}

// fused per-frame robustness -> R: k_rob_frame with one channel and the same-size statistics map
__global__ void __launch_bounds__(256) k_mono_rob_frame(const float* __restrict__ cm, const float* __restrict__ rmean,
                                                         const float* __restrict__ ssq,
                                                         const float2* __restrict__ flow, int nx, int ts,
                                                         const float* __restrict__ S, const double* __restrict__ difc,
                                                         int ncurve, double t, float* __restrict__ R, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int tix = x / ts, tiy = y / ts;
    const float2 f = flow[(size_t)tiy * nx + tix];
    float cmu[1];
    if (!dodgson_sample<1>(cm, H, W, y, x, (double)f.x, (double)f.y, cmu)) cmu[0] = INFINITY;
    const size_t o = (size_t)y * W + x;
    const float b = rmean[o];
    const float dp = fabsf(b - cmu[0]);
    int id = 0;
    const double bb = 1000.0 * (double)b;
    if (isfinite(bb)) id = clampi((int)rint(bb), 0, ncurve - 1);
    const double d_t = difc[id];
    const float dp2f = dp * dp;
    const double dp2 = (double)dp2f;
    const double shrink = dp2 / (dp2 + d_t * d_t);
    const double d_sq = 0.0 + dp2 * shrink * shrink;
    const float dsf = (float)d_sq, ssf = ssq[o];
    const float e = expf(-dsf / ssf);
    double v = (double)(S[(size_t)tiy * nx + tix] * e) - t;
    v = v > 0.0 ? v : 0.0;  // NaN -> 0
    v = v < 1.0 ? v : 1.0;
    R[o] = (float)v;
}

// The same with the float32 arithmetic of the Bayer kernels (see k_rob_frame_tile: exact integer / predicate geometry,
// float32 Dodgson weights, v_rcp_f32 / v_exp_f32 tail; |dR| <= 1e-4) and a thread per 4 horizontally adjacent pixels
// (x0 % 4 == 0, ts % 4 == 0: one flow tile).  Pixels k and k + 2 lie one guide pixel apart with the same sub-pixel
// phase and share the weights; the two phases' 3 x 4 windows overlap in a 3 x 5 window read once from L1 / L2 (the
// guide positions of a 64 x 16 pixel workgroup cover 35 x 11 values).  The reference-frame planes and R move as
// 16-byte vectors.  The float64 kernel above took 248 us per 12 MP frame (VALU: float64 Dodgson weights, two float64
// divisions and an expf per pixel, 9 dword loads); it remains the fall-back for odd widths / tile sizes.
__global__ void __launch_bounds__(256) k_mono_rob_frame4(const float* __restrict__ cm, const float* __restrict__ rmean,
                                                          const float* __restrict__ ssq,
                                                          const float2* __restrict__ flow, int nx, int ts,
                                                          const float* __restrict__ S, const double* __restrict__ difc,
                                                          int ncurve, float tf, float* __restrict__ R, int H, int W) {
    const int x0 = (blockIdx.x * 16 + (threadIdx.x & 15)) * 4, y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x0 >= W || y >= H) return;
    const int tix = x0 / ts, tiy = y / ts;
    const float2 f = flow[(size_t)tiy * nx + tix];
    const float Sv = S[(size_t)tiy * nx + tix];
    const RobAxis ay = rob_axis(f.y), ax = rob_axis(f.x);
    const size_t o = (size_t)y * W + x0;
    const float4 rb4 = *reinterpret_cast<const float4*>(rmean + o);
    const float4 ss4 = *reinterpret_cast<const float4*>(ssq + o);
    const float rbk[4] = {rb4.x, rb4.y, rb4.z, rb4.w}, ssk[4] = {ss4.x, ss4.y, ss4.z, ss4.w};
    int cy, cx[4];
    float ry, rx[4], wyv[3];
    bool inx[4];
    const bool iny = rob_centre(ay, y, H, cy, ry);
    dodgson3(ry, cy, H, wyv);
#pragma unroll
    for (int k = 0; k < 4; ++k) inx[k] = rob_centre(ax, x0 + k, W, cx[k], rx[k]);
    const float* row[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) row[i] = cm + (size_t)clampi(cy - 1 + i, 0, H - 1) * W;
    float cmu[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    // no clamped tap in x for any of the 4 pixels, and no round-half-even tie (there pixel k + 2 rounds the other way)
    if (iny && inx[0] && inx[1] && !ax.eq && cx[0] >= 1 && cx[1] + 2 <= W - 1) {
        const int c0 = cx[0] - 1;
        const bool d = cx[1] != cx[0];  // phase 1 starts at c0 or c0 + 1
        float g[3][5];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) g[i][j] = row[i][min(c0 + j, W - 1)];
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            float wxv[3], w[3][3], wacc = 0.f;
            dodgson3(rx[ph], cx[ph], W, wxv);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    w[i][j] = wyv[i] * wxv[j];
                    wacc += w[i][j];
                }
            const float iw = __builtin_amdgcn_rcpf(wacc);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float b = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float t_ = (ph && d) ? g[i][j + q + 1] : g[i][j + q];
                        b = fmaf(t_, w[i][j], b);
                    }
                cmu[ph + 2 * q] = b * iw;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!(iny && inx[k])) continue;
            float wxv[3], b = 0.f, wacc = 0.f;
            dodgson3(rx[k], cx[k], W, wxv);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float w = wyv[i] * wxv[j];
                    b = fmaf(row[i][clampi(cx[k] - 1 + j, 0, W - 1)], w, b);
                    wacc += w;
                }
            cmu[k] = b * __builtin_amdgcn_rcpf(wacc);
        }
    }
    float out[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int id = 0;
        const double bb = 1000.0 * (double)rbk[k];  // the curve index in float64 like the reference (robustness.py:517)
        if (isfinite(bb)) id = clampi((int)rint(bb), 0, ncurve - 1);
        const float d_t = (float)difc[id];
        const float dp = fabsf(rbk[k] - cmu[k]);
        const float dp2 = dp * dp;
        const float shrink = dp2 * __builtin_amdgcn_rcpf(dp2 + d_t * d_t);
        const float d_sq = dp2 * shrink * shrink;
        const float e = __builtin_amdgcn_exp2f((-d_sq * __builtin_amdgcn_rcpf(ssk[k])) * 1.44269504088896341f);
        out[k] = __builtin_amdgcn_fmed3f(Sv * e - tf, 0.f, 1.f);  // NaN -> 0 like the reference's clamp
    }
    *reinterpret_cast<float4*>(R + o) = make_float4(out[0], out[1], out[2], out[3]);
}

extern "C" int hhsr_mono_rob_frame(const float* comp_means, int H, int W, const float* ref_means,
                                   const float* sigma_sq, const float* flow, int ny, int nx, int ts, const float* S,
                                   const double* diff_curve, int ncurve, double t, float* R, void* stream) {
    HHSR_ARG(comp_means && ref_means && sigma_sq && flow && S && diff_curve && R);
    HHSR_ARG(H > 0 && W > 0 && ts > 0 && ncurve > 0 && (int64_t)ny * ts >= H && (int64_t)nx * ts >= W);
    const bool vec = W % 4 == 0 && ts % 4 == 0 &&
                     (((uintptr_t)ref_means | (uintptr_t)sigma_sq | (uintptr_t)R) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(k_mono_rob_frame4, dim3(hhsr_cdiv(W, 64), hhsr_cdiv(H, 16)), dim3(256), 0,
                           (hipStream_t)stream, comp_means, ref_means, sigma_sq,
                           reinterpret_cast<const float2*>(flow), nx, ts, S, diff_curve, ncurve, (float)t, R, H, W);
    else
        hipLaunchKernelGGL(k_mono_rob_frame, dim3(hhsr_cdiv(W, 64), hhsr_cdiv(H, 4)), dim3(256), 0,
                           (hipStream_t)stream, comp_means, ref_means, sigma_sq,
                           reinterpret_cast<const float2*>(flow), nx, ts, S, diff_curve, ncurve, t, R, H, W);
    HHSR_LAUNCHED();
}

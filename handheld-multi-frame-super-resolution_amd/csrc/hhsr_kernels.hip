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


struct CovParams {
    double alpha, beta, k_detail, k_denoise, D_th, D_tr, k_stretch, k_shrink;
    int law;
    double r_D_tr, inv_k_shrink;  // host-computed RN(1 / D_tr), 1.0 / k_shrink
};

// a / b for a divisor with a precomputed r = RN(1 / b): q = RN(a r), one FMA residual correction.  Correctly
// rounded (Markstein); checked against true division on 4e5 float32-origin operands for b = 9.  Replaces the
// ~35-instruction IEEE float64 division sequence by 3 instructions.
__device__ __forceinline__ double div_by(double a, double b, double r) {
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// sqrt of a non-negative double in the normal range (no sub-normal / infinite argument handling: the library sqrt
// spends ~10 of its ~28 instructions on scaling by 2^+-256 and class tests).  v_rsq_f64 seed, one Goldschmidt
// iteration and one residual correction: error <= 1 ulp (the library: correctly rounded); the callers round the
// result to float32, where a 1-ulp float64 difference changes the float32 value with probability ~2^-29.
#ifndef HHSR_FAST_SQRT
#define HHSR_FAST_SQRT 1
#endif
__device__ __forceinline__ double sqrt_pos(double t) {
#if HHSR_FAST_SQRT
    const double y = __builtin_amdgcn_rsq(t);
    double g = t * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, t);
    return fma(d, h, g);  // (t = 0: rsq = inf -> NaN; gat1 keeps its argument >= 1e-300, the eigenvalue root guards itself)
#else
    return sqrt(t);
#endif
}

// float32 quotient of operands in the normal range: v_rcp_f32 (1 ulp) + one residual correction — the correctly
// rounded quotient except in rare double-rounding cases (the IEEE expansion spends ~10 instructions, most of them on
// operand scaling); 0 / 0 and x / NaN stay NaN (D10), a zero divisor with a non-zero dividend does not occur here.
#ifndef HHSR_FAST_DIV
#define HHSR_FAST_DIV 1
#endif
__device__ __forceinline__ float div_f32(float x, float y) {
#if HHSR_FAST_DIV
    const float r = __builtin_amdgcn_rcpf(y), q = x * r;
    return fmaf(fmaf(-y, q, x), r, q);
#else
    return x / y;
#endif
}

__device__ __forceinline__ float gat1(float v, double alpha, double c0, double two_over_alpha) {
    // VST = alpha*I + 3/8*alpha^2 + beta ; max(0, .) ; 2/alpha * sqrt(.)   (utils_image.py:167-170)
    // max(0, .) as max(1e-300, .): the float32 result of an argument <= 0 is 0 either way (2 / alpha * 1e-150 rounds to 0)
    // and sqrt_pos needs no zero test (a float64 compare and two selects per pixel, 12 pixels per thread)
    const double t = fmax(alpha * (double)v + c0, 1.0e-300);
    return (float)(two_over_alpha * sqrt_pos(t));
}

// Per-quad kernel covariance from the variance-stabilised quad means `sg` (LDS tile with a one-quad halo,
// pitch SGP; (ly, lx) = the quad's position inside the tile without the halo).
template <int SGP>
__device__ __forceinline__ float4 quad_cov(const float* __restrict__ sg, int ly, int lx, int qy, int qx, int gh, int gw,
                                           const CovParams& P) {
    // structure tensor over the gradient samples (y-1..y, x-1..x) that exist in the [gh-1][gw-1] grid
    float T00 = 0.f, T01 = 0.f, T11 = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gy_ = qy - 1 + i, gx_ = qx - 1 + j;
            if (gy_ >= 0 && gy_ < gh - 1 && gx_ >= 0 && gx_ < gw - 1) {
                const float g00 = sg[(ly + i) * SGP + lx + j], g01 = sg[(ly + i) * SGP + lx + j + 1];
                const float g10 = sg[(ly + i + 1) * SGP + lx + j], g11 = sg[(ly + i + 1) * SGP + lx + j + 1];
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
    const double sq = delta > 0.0 ? sqrt_pos(delta) : 0.0;
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
            e1x = div_f32(a0, nrm);
            e1y = div_f32(a1, nrm);
            const float sgn = copysignf(1.f, e1x);
            e2y = fabsf(e1x);
            e2x = -e1y * sgn;
        }
    }
    // k1, k2 (kernels.py:195-243): A, D float64 from float32 square roots; k stored float32
    const double A = 1.0 + (double)sqrtf(div_f32(l1 - l2, l1 + l2));
    double D = 1.0 - div_by((double)sqrtf(l1), P.D_tr, P.r_D_tr) + P.D_th;
    D = D > 0.0 ? D : 0.0;  // clamp with Python max/min semantics (NaN -> 0)
    D = D < 1.0 ? D : 1.0;
    double k1d, k2d;
    if (P.law == 0) {  // hard_threshold; a NaN anisotropy falls into the else branch
        if (A > 1.95) {
            k1d = P.inv_k_shrink;
            k2d = P.k_stretch;
        } else {
            k1d = 1.0;
            k2d = 1.0;
        }
    } else {  // linear
        k1d = 1.0 + A / 2.0 * (P.inv_k_shrink - 1.0);
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
    return o;
}

// ---- fused per-frame pass over the raw image -----------------------------------------------------------------
// Both per-frame consumers of the raw Bayer image at quad resolution — the guide-image local statistics of the
// robustness (Alg. 7-8) and the kernel covariances (Alg. 5) — work on a quad tile with a one-quad halo.  One
// kernel stages the tile once (4 B/pixel read instead of 8), each thread owns two quads (rows ly and ly + 8 of a
// 32 x 16 tile: 19 % halo overhead instead of 33 %), and outputs that the caller does not need (the variances of
// comp frames) are not written.  The stand-alone entry points hhsr_rob_stats / hhsr_cov_from_raw run the same
// kernel with one half compiled out.
constexpr int FS_TX = 32, FS_TY = 16, FS_P = FS_TX + 2 + 1, FS_N = (FS_TY + 2) * (FS_TX + 2);
constexpr int FS_IT = (FS_N + 255) / 256;

struct FsCfa {
    uint8_t c[4];
};
struct FsFrames {  // blockIdx.z = frame of the batch
    const float* raw[HHSR_MAX_BATCH];
    float* means[HHSR_MAX_BATCH];   // [3][gh][gw] or NULL
    float* vars[HHSR_MAX_BATCH];    // [3][gh][gw] or NULL
    float4* covs[HHSR_MAX_BATCH];   // [gh][gw] or NULL
};

struct FrameStatsArgs {
    const float* raw;
    int pitch, gh, gw;
    FsCfa cfa;
    double rwb[3];  // 1 / white balance
    double rwbk[4]; // ... of the colour at quad position k
    int pat;        // 0 RGGB, 1 BGGR, 2 GRBG, 3 GBRG (compile-time channel routing per case); -1: any other 2 x 2 pattern
    int unit_wb;    // white balance (1, 1, 1): the guide channels are the raw values (float32-exact fast path)
    float* means;   // [3][gh][gw] or NULL
    float* vars;    // [3][gh][gw] or NULL
    float4* covs;   // [gh][gw] or NULL
    CovParams P;
};

template <bool STATS, bool COV>
__global__ void __launch_bounds__(256) k_frame_stats(FrameStatsArgs A, FsFrames fr) {
    A.raw = fr.raw[blockIdx.z];
    A.means = fr.means[blockIdx.z];
    A.vars = fr.vars[blockIdx.z];
    A.covs = fr.covs[blockIdx.z];
    __shared__ float s_ch[STATS ? 3 : 1][STATS ? FS_TY + 2 : 1][FS_P];
    __shared__ float s_g[COV ? (FS_TY + 2) * FS_P : 1];
    const int bid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);  // bands of tile rows per XCD
    const int qx0 = (bid % gridDim.x) * FS_TX, qy0 = (bid / gridDim.x) * FS_TY;
    const int gh = A.gh, gw = A.gw;
    const double c0 = 3.0 / 8.0 * A.P.alpha * A.P.alpha + A.P.beta;
    const double toa = 2.0 / A.P.alpha;
    // all loads of the tile first, then the float64 maths (a load -> compute -> store loop would serialise on
    // memory latency); out-of-image quads read the clamped quad (the statistics' border rule) and are zeroed for
    // the covariance tile (its border rule)
    float2 va[FS_IT], vb[FS_IT];
#pragma unroll
    for (int u = 0; u < FS_IT; ++u) {
        const int p = min(threadIdx.x + 256 * u, FS_N - 1);
        const int i = p / (FS_TX + 2), j = p - i * (FS_TX + 2);
        const int qy = clampi(qy0 + i - 1, 0, gh - 1), qx = clampi(qx0 + j - 1, 0, gw - 1);
        va[u] = *reinterpret_cast<const float2*>(A.raw + (size_t)(2 * qy) * A.pitch + 2 * qx);
        vb[u] = *reinterpret_cast<const float2*>(A.raw + (size_t)(2 * qy + 1) * A.pitch + 2 * qx);
    }
#pragma unroll
    for (int u = 0; u < FS_IT; ++u) {
        const int p = threadIdx.x + 256 * u;
        if (p < FS_N) {
            const int i = p / (FS_TX + 2), j = p - i * (FS_TX + 2);
            const float v[4] = {va[u].x, va[u].y, vb[u].x, vb[u].y};
            if (STATS) {
                // robustness.py:215-225: raw / wb[c] in float64; the two greens are added in (i, j) order and halved.
                // The Bayer patterns route the four positions to their channels at compile time (a run-time colour
                // index costs ~9 selects per pixel: 108 of this kernel's 173 v_cndmask)
                float ch[3] = {0.f, 0.f, 0.f};
                const double x0 = (double)v[0] * A.rwbk[0], x1 = (double)v[1] * A.rwbk[1];
                const double x2 = (double)v[2] * A.rwbk[2], x3 = (double)v[3] * A.rwbk[3];
                if (A.unit_wb && A.pat >= 0) {
                    // white balance (1, 1, 1): raw / 1.0 is the raw value, and the float64 green mean ((0 + a) + b) / 2
                    // rounded to float32 IS the float32 sum halved — for operands whose exponents differ by at most 29 bits
                    // (then a + b is exact in float64) and whose sum is a normal float32 (then halving is exact): normalised
                    // sensor data in [0, 1] with values >= 2^-100 or exactly 0; beyond that the float64 route rounds twice
                    // and may differ by one ulp.  No float64 instruction left in this block (11 per staged quad)
                    const float g01 = 0.5f * (v[1] + v[2]), g03 = 0.5f * (v[0] + v[3]);
                    ch[1] = A.pat <= 1 ? g01 : g03;
                    ch[0] = A.pat == 0 ? v[0] : A.pat == 1 ? v[3] : A.pat == 2 ? v[1] : v[2];
                    ch[2] = A.pat == 0 ? v[3] : A.pat == 1 ? v[0] : A.pat == 2 ? v[2] : v[1];
                } else if (A.pat == 0) {  // R G / G B
                    ch[0] = (float)x0; ch[1] = (float)(((0.0 + x1) + x2) / 2.0); ch[2] = (float)x3;
                } else if (A.pat == 1) {  // B G / G R
                    ch[2] = (float)x0; ch[1] = (float)(((0.0 + x1) + x2) / 2.0); ch[0] = (float)x3;
                } else if (A.pat == 2) {  // G R / B G
                    ch[0] = (float)x1; ch[1] = (float)(((0.0 + x0) + x3) / 2.0); ch[2] = (float)x2;
                } else if (A.pat == 3) {  // G B / R G
                    ch[2] = (float)x1; ch[1] = (float)(((0.0 + x0) + x3) / 2.0); ch[0] = (float)x2;
                } else {
                    double g = 0.0;
                    const double x[4] = {x0, x1, x2, x3};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = A.cfa.c[k];
                        if (c == 1) g += x[k];
                        else ch[c] = (float)x[k];
                    }
                    ch[1] = (float)(g / 2.0);
                }
                s_ch[0][i][j] = ch[0];
                s_ch[1][i][j] = ch[1];
                s_ch[2][i][j] = ch[2];
            }
            if (COV) {
                const int qy = qy0 + i - 1, qx = qx0 + j - 1;
                float g = 0.f;
                if (qy >= 0 && qy < gh && qx >= 0 && qx < gw) {
                    // decimate: float64 sum of the four float32 VST values, /4 (utils_image.py:346-357)
                    const double s = (double)gat1(v[0], A.P.alpha, c0, toa) + (double)gat1(v[1], A.P.alpha, c0, toa) +
                                     (double)gat1(v[2], A.P.alpha, c0, toa) + (double)gat1(v[3], A.P.alpha, c0, toa);
                    g = (float)(s / 4.0);
                }
                s_g[i * FS_P + j] = g;
            }
        }
    }
    __syncthreads();
    const int lx = threadIdx.x % FS_TX, ly0 = threadIdx.x / FS_TX;
    const int gx = qx0 + lx;
    if (gx >= gw) return;
    const size_t plane = (size_t)gh * gw;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ly = ly0 + 8 * h, gy = qy0 + ly;
        if (gy >= gh) break;
        const size_t o = (size_t)gy * gw + gx;
        if (STATS) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s0 = 0.f, s1 = 0.f;  // float32 running sums in (i, j) order (robustness.py:280-288)
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float v = s_ch[c][ly + i][lx + j];
                        s0 += v;
                        s1 += v * v;
                    }
                // mean = float32(float64(s0) / 9): the correctly rounded float32 quotient (a float32 over 9 never lies within
                // 2^-28 relative of a float32 rounding boundary, so rounding through float64 changes nothing), which
                // q = s0 r, q + (s0 - 9 q) r with r = RN(1 / 9) delivers in float32 — checked exhaustively over all 2^23
                // mantissas (Markstein) at exponents where the residual s0 - 9 q is a normal number (|s0| >= 2^-100: sums of
                // nine normalised samples, or exactly 0); a tinier sum may misround by one ulp.  The variance keeps the
                // float64 mean
                const float r9 = 1.0f / 9.0f, q9 = s0 * r9;
                A.means[c * plane + o] = fmaf(fmaf(-9.0f, q9, s0), r9, q9);
                if (A.vars) {
                    const double m = div_by((double)s0, 9.0, 1.0 / 9.0);
                    A.vars[c * plane + o] = (float)(div_by((double)s1, 9.0, 1.0 / 9.0) - m * m);
                }
            }
        }
        if (COV) A.covs[o] = quad_cov<FS_P>(s_g, ly, lx, gy, gx, gh, gw, A.P);
    }
}

static int frame_stats_launch_batch(const float* const* raws, int n_frames, int H, int W, int pitch, const uint8_t* cfa,
                                    const double* wb, float* const* means, float* const* vars, float* const* covs,
                                    const CovParams& P, void* stream) {
    FrameStatsArgs A;
    A.raw = nullptr; A.pitch = pitch; A.gh = H / 2; A.gw = W / 2;
    for (int k = 0; k < 4; ++k) A.cfa.c[k] = cfa ? cfa[k] : 0;
    for (int k = 0; k < 3; ++k) A.rwb[k] = wb ? 1.0 / wb[k] : 1.0;
    for (int k = 0; k < 4; ++k) A.rwbk[k] = A.rwb[A.cfa.c[k]];
    const int code = A.cfa.c[0] * 27 + A.cfa.c[1] * 9 + A.cfa.c[2] * 3 + A.cfa.c[3];
    A.pat = code == 0 * 27 + 1 * 9 + 1 * 3 + 2 ? 0 : code == 2 * 27 + 1 * 9 + 1 * 3 + 0 ? 1
          : code == 1 * 27 + 0 * 9 + 2 * 3 + 1 ? 2 : code == 1 * 27 + 2 * 9 + 0 * 3 + 1 ? 3 : -1;
    A.unit_wb = A.rwb[0] == 1.0 && A.rwb[1] == 1.0 && A.rwb[2] == 1.0;
    A.means = nullptr; A.vars = nullptr; A.covs = nullptr; A.P = P;
    A.P.r_D_tr = 1.0 / P.D_tr;
    A.P.inv_k_shrink = 1.0 / P.k_shrink;
    hipStream_t s = (hipStream_t)stream;
    for (int n0 = 0; n0 < n_frames; n0 += HHSR_MAX_BATCH) {
        const int nb = n_frames - n0 < HHSR_MAX_BATCH ? n_frames - n0 : HHSR_MAX_BATCH;
        FsFrames fr;
        for (int k = 0; k < HHSR_MAX_BATCH; ++k) {
            const int n = n0 + (k < nb ? k : 0);
            fr.raw[k] = raws[n];
            fr.means[k] = means ? means[n] : nullptr;
            fr.vars[k] = vars ? vars[n] : nullptr;
            fr.covs[k] = covs ? reinterpret_cast<float4*>(covs[n]) : nullptr;
        }
        const dim3 grid(hhsr_cdiv(A.gw, FS_TX), hhsr_cdiv(A.gh, FS_TY), nb), block(256);
        if (means && covs) hipLaunchKernelGGL((k_frame_stats<true, true>), grid, block, 0, s, A, fr);
        else if (means) hipLaunchKernelGGL((k_frame_stats<true, false>), grid, block, 0, s, A, fr);
        else hipLaunchKernelGGL((k_frame_stats<false, true>), grid, block, 0, s, A, fr);
    }
    return hhsr_launch_status("hhsr_frame_stats");
}

static int frame_stats_launch(const float* raw, int H, int W, int pitch, const uint8_t* cfa, const double* wb,
                              float* means, float* vars, float* covs, const CovParams& P, void* stream) {
    return frame_stats_launch_batch(&raw, 1, H, W, pitch, cfa, wb, means ? &means : nullptr, vars ? &vars : nullptr,
                                    covs ? &covs : nullptr, P, stream);
}

extern "C" int hhsr_cov_from_raw(const float* raw, int H, int W, int pitch, float* covs, double alpha, double beta,
                                 double k_detail, double k_denoise, double D_th, double D_tr, double k_stretch,
                                 double k_shrink, int law, void* stream) {
    HHSR_ARG(raw && covs && H >= 2 && W >= 2 && pitch >= W);
    HHSR_ARG((pitch & 1) == 0 && ((uintptr_t)raw & 7) == 0 && ((uintptr_t)covs & 15) == 0);
    HHSR_ARG(alpha > 0.0);  // utils_image.py:141: the VST is ill-defined otherwise
    HHSR_ARG(law == 0 || law == 1);
    CovParams P{alpha, beta, k_detail, k_denoise, D_th, D_tr, k_stretch, k_shrink, law};
    return frame_stats_launch(raw, H, W, pitch, nullptr, nullptr, nullptr, nullptr, covs, P, stream);
}

extern "C" int hhsr_rob_stats(const float* raw, int H, int W, int pitch, const uint8_t cfa[4], const double* wb,
                              float* means, float* vars, void* stream) {
    HHSR_ARG(raw && cfa && wb && means && H >= 2 && W >= 2 && pitch >= W);
    HHSR_ARG((pitch & 1) == 0 && ((uintptr_t)raw & 7) == 0);
    for (int k = 0; k < 4; ++k) HHSR_ARG(cfa[k] <= 2);
    for (int k = 0; k < 3; ++k) HHSR_ARG(wb[k] != 0.0);
    CovParams P{1.0, 0.0, 0, 0, 0, 1.0, 0, 1.0, 0};
    return frame_stats_launch(raw, H, W, pitch, cfa, wb, means, vars, nullptr, P, stream);
}

extern "C" int hhsr_frame_stats(const float* raw, int H, int W, int pitch, const uint8_t cfa[4], const double* wb,
                                float* means, float* vars, float* covs, double alpha, double beta, double k_detail,
                                double k_denoise, double D_th, double D_tr, double k_stretch, double k_shrink, int law,
                                void* stream) {
    HHSR_ARG(raw && cfa && wb && means && covs && H >= 2 && W >= 2 && pitch >= W);
    HHSR_ARG((pitch & 1) == 0 && ((uintptr_t)raw & 7) == 0 && ((uintptr_t)covs & 15) == 0);
    HHSR_ARG(alpha > 0.0 && (law == 0 || law == 1));
    for (int k = 0; k < 4; ++k) HHSR_ARG(cfa[k] <= 2);
    for (int k = 0; k < 3; ++k) HHSR_ARG(wb[k] != 0.0);
    CovParams P{alpha, beta, k_detail, k_denoise, D_th, D_tr, k_stretch, k_shrink, law};
    return frame_stats_launch(raw, H, W, pitch, cfa, wb, means, vars, covs, P, stream);
}

extern "C" int hhsr_frame_stats_batch(const float* const* raws, int n_frames, int H, int W, int pitch, const uint8_t cfa[4],
                                      const double* wb, float* const* means, float* const* covs, double alpha,
                                      double beta, double k_detail, double k_denoise, double D_th, double D_tr,
                                      double k_stretch, double k_shrink, int law, void* stream) {
    HHSR_ARG(raws && cfa && wb && means && covs && n_frames >= 0 && H >= 2 && W >= 2 && pitch >= W && (pitch & 1) == 0);
    for (int n = 0; n < n_frames; ++n)
        HHSR_ARG(raws[n] && means[n] && covs[n] && ((uintptr_t)raws[n] & 7) == 0 && ((uintptr_t)covs[n] & 15) == 0);
    HHSR_ARG(alpha > 0.0 && (law == 0 || law == 1));
    for (int k = 0; k < 4; ++k) HHSR_ARG(cfa[k] <= 2);
    for (int k = 0; k < 3; ++k) HHSR_ARG(wb[k] != 0.0);
    CovParams P{alpha, beta, k_detail, k_denoise, D_th, D_tr, k_stretch, k_shrink, law};
    return frame_stats_launch_batch(raws, n_frames, H, W, pitch, cfa, wb, means, nullptr, covs, P, stream);
}

// ---- monochrome sensors (`mode: grey`) ---------------------------------------------------------------------------
// The frame is its own guide image (one channel, white balance not involved: robustness.py:62-66) and the kernel
// covariances are estimated per PIXEL from the stabilised frame itself (kernels.py:83-87): the same tile kernel with
// pixels in the place of Bayer quads — 32 x 16 pixels per workgroup with a one-pixel halo, two pixels per thread.
struct MonoStatsArgs {
    const float* raw;
    int pitch, H, W;
    float* means;  // [H][W] or NULL
    float* vars;   // [H][W] or NULL
    float4* covs;  // [H][W] or NULL
    CovParams P;
};

template <bool STATS, bool COV>
__global__ void __launch_bounds__(256) k_mono_stats(MonoStatsArgs A) {
    __shared__ float s_v[STATS ? FS_TY + 2 : 1][FS_P];
    __shared__ float s_g[COV ? (FS_TY + 2) * FS_P : 1];
    const int bid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int x0 = (bid % gridDim.x) * FS_TX, y0 = (bid / gridDim.x) * FS_TY;
    const int H = A.H, W = A.W;
    const double c0 = 3.0 / 8.0 * A.P.alpha * A.P.alpha + A.P.beta;
    const double toa = 2.0 / A.P.alpha;
    float v[FS_IT];
#pragma unroll
    for (int u = 0; u < FS_IT; ++u) {  // clamped pixel = the statistics' border rule (robustness.py:281-282)
        const int p = min(threadIdx.x + 256 * u, FS_N - 1);
        const int i = p / (FS_TX + 2), j = p - i * (FS_TX + 2);
        v[u] = A.raw[(size_t)clampi(y0 + i - 1, 0, H - 1) * A.pitch + clampi(x0 + j - 1, 0, W - 1)];
    }
#pragma unroll
    for (int u = 0; u < FS_IT; ++u) {
        const int p = threadIdx.x + 256 * u;
        if (p < FS_N) {
            const int i = p / (FS_TX + 2), j = p - i * (FS_TX + 2);
            if (STATS) s_v[i][j] = v[u];
            if (COV) {  // zero outside the frame: quad_cov only reads gradient samples that exist
                const int y = y0 + i - 1, x = x0 + j - 1;
                s_g[i * FS_P + j] = (y >= 0 && y < H && x >= 0 && x < W) ? gat1(v[u], A.P.alpha, c0, toa) : 0.f;
            }
        }
    }
    __syncthreads();
    const int lx = threadIdx.x % FS_TX, ly0 = threadIdx.x / FS_TX;
    const int gx = x0 + lx;
    if (gx >= W) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ly = ly0 + 8 * h, gy = y0 + ly;
        if (gy >= H) break;
        const size_t o = (size_t)gy * W + gx;
        if (STATS) {
            float s0 = 0.f, s1 = 0.f;  // float32 running sums in (i, j) order (robustness.py:280-288)
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float t = s_v[ly + i][lx + j];
                    s0 += t;
                    s1 += t * t;
                }
            const float r9 = 1.0f / 9.0f, q9 = s0 * r9;  // (see k_frame_stats)
            A.means[o] = fmaf(fmaf(-9.0f, q9, s0), r9, q9);
            if (A.vars) {
                const double m = div_by((double)s0, 9.0, 1.0 / 9.0);
                A.vars[o] = (float)(div_by((double)s1, 9.0, 1.0 / 9.0) - m * m);
            }
        }
        if (COV) A.covs[o] = quad_cov<FS_P>(s_g, ly, lx, gy, gx, H, W, A.P);
    }
}

extern "C" int hhsr_mono_frame_stats(const float* raw, int H, int W, int pitch, float* means, float* vars, float* covs,
                                     double alpha, double beta, double k_detail, double k_denoise, double D_th,
                                     double D_tr, double k_stretch, double k_shrink, int law, void* stream) {
    HHSR_ARG(raw && (means || covs) && H >= 2 && W >= 2 && pitch >= W);
    HHSR_ARG(!vars || means);
    HHSR_ARG(!covs || (((uintptr_t)covs & 15) == 0 && alpha > 0.0 && (law == 0 || law == 1)));
    MonoStatsArgs A;
    A.raw = raw; A.pitch = pitch; A.H = H; A.W = W;
    A.means = means; A.vars = vars; A.covs = reinterpret_cast<float4*>(covs);
    A.P = covs ? CovParams{alpha, beta, k_detail, k_denoise, D_th, D_tr, k_stretch, k_shrink, law}
               : CovParams{1.0, 0.0, 0, 0, 0, 1.0, 0, 1.0, 0};
    A.P.r_D_tr = 1.0 / A.P.D_tr;
    A.P.inv_k_shrink = 1.0 / A.P.k_shrink;
    const dim3 grid(hhsr_cdiv(W, FS_TX), hhsr_cdiv(H, FS_TY)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (means && covs) hipLaunchKernelGGL((k_mono_stats<true, true>), grid, block, 0, s, A);
    else if (means) hipLaunchKernelGGL((k_mono_stats<true, false>), grid, block, 0, s, A);
    else hipLaunchKernelGGL((k_mono_stats<false, true>), grid, block, 0, s, A);
    return hhsr_launch_status("hhsr_mono_frame_stats");
}

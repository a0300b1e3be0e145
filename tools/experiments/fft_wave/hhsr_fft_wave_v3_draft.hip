// Wave-synchronous in-LDS FFT low-pass for the grey image (Alg. 3, reference utils_image.py:82-100): the same three
// phases as hhsr_fft.hip (rows forward -> columns forward / mask / inverse -> rows inverse) with ONE WAVE per 1-D
// transform and a radix schedule fixed at compile time.
//
// Why: the workgroup-synchronous kernels (hhsr_fft.hip: 256 / 512 threads per transform, two __syncthreads() per pass)
// issue ~760 VALU instructions per row and wave for ~340 of butterfly arithmetic (run-time radices: every LDS address is
// computed, integer divisions by float tricks) and park 45 % of their wave time at barriers.  Here
//   * a wave owns its transform from the global load to the global store: the LDS operations of one wave execute in
//     program order, so the exchange between two passes needs no s_barrier and no s_waitcnt beyond the data dependences
//     — only the compiler's schedule is fenced (wave_fence());
//   * N, the radices and every sub-transform length are template constants: butterfly r of a pass sits at an IMMEDIATE
//     offset of the lane's base address, q = j / Ns and k = j % Ns are divisions by constants;
//   * the first pass takes its inputs straight from global memory (lane j reads x[j + r L]: consecutive lanes, consecutive
//     addresses) and the last pass of an inverse stores straight to global memory — no staging copy through LDS;
//   * a transform lives in the wave's REGISTERS between two passes; LDS is only the exchange medium, and the exchange runs
//     in two halves — all real parts out and back in, then all imaginary parts — through a float buffer of N x 4 bytes:
//     8 / 12 KB per wave instead of 16 / 24, i.e. 16 / 10 waves per CU instead of 8 / 4.  One wave issues a VALU instruction
//     every ~5.5 cycles whatever its ILP (tools/ubench/valu_occupancy): with 2 waves per SIMD the first designs of this file
//     (profiles/r06_fft_wave_v1_notes.txt) ran slower than the workgroup kernels; occupancy, not prefetch distance, is
//     what hides a free-running wave's LDS and memory round trips.
// The butterflies are those of hhsr_fft.hip (hhsr_fft_bfly.h).  The column kernel runs the workgroup kernel's schedule on
// its tables, the row kernels a schedule that fills 64 lanes (own tables); both agree with the workgroup kernels within
// float32 rounding (tests/test_hip_parity.py::test_grey_wave_kernels_equal_workgroup_kernels).
#include "hhsr_common.h"
#include "hhsr_fft.h"
#include "hhsr_fft_bfly.h"
#include <math.h>
#include <vector>

namespace {

// ---- static plans ---------------------------------------------------------------------------------------------------
template <int N_, int... RS>
struct WPlan {
    static constexpr int N = N_;
    static constexpr int NP = (int)sizeof...(RS);
    static constexpr int R[NP] = {RS...};
    static constexpr int ns(int p) {  // sub-transform length before pass p
        int s = 1;
        for (int i = 0; i < p; ++i) s *= R[i];
        return s;
    }
    static constexpr bool pw(int p) {  // pass p keeps w^k only (= pass_twiddles() of hhsr_fft.hip with the default pow_min)
        return HHSR_FFT_POW_MIN > 0 && R[p] <= HHSR_FFT_POW_RMAX && (R[p] - 1) * ns(p) > HHSR_FFT_POW_MIN;
    }
    static constexpr int toff(int p) {  // offset of pass p's table in the concatenated twiddle table
        int o = 0;
        for (int i = 0; i < p; ++i) o += pw(i) ? ns(i) : (R[i] - 1) * ns(i);
        return o;
    }
    static constexpr int twlen = toff(NP);
    static constexpr int maxvals() {  // float2 registers one wave holds in its widest pass
        int m = 0;
        for (int i = 0; i < NP; ++i) {
            const int v = ((N / R[i] + 63) / 64) * R[i];
            m = v > m ? v : m;
        }
        return m;
    }
};

// The compiler's schedule only: all LDS reads of a phase are issued before the first write of the next (and the other
// way round).  The hardware keeps a wave's LDS operations in order; no instruction is generated.
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The lane id as a value the compiler cannot trace: every index of a pass (j / Ns, j % Ns, the LDS byte addresses) is a
// function of the lane alone, i.e. invariant over the kernel's transform loop — hoisted out of it they are ~70 values that
// stay live across the whole loop body and get spilled.  Recomputing them per pass is a handful of instructions.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Butterfly j = lane + 64 it of a radix-R pass over N points (L = N / R butterflies, RO rounds of 64): its inputs are
// x[j + r L], its outputs x[(j / Ns) R Ns + j % Ns + r Ns] (Stockham autosort).  Register layout: round it, leg r at v[it R + r].
template <int N, int R>
struct WGeo {
    static constexpr int L = N / R, RO = (L + 63) / 64;
    static __device__ __forceinline__ bool on(int it, int j) { return (it + 1) * 64 <= L || j < L; }
};

// inputs of a pass from global memory (or any float2 source)
template <int N, int R, typename Src>
__device__ __forceinline__ void w_load(Src src, int lane, float2* v) {
    using G = WGeo<N, R>;
#pragma unroll
    for (int it = 0; it < G::RO; ++it) {
        const int j = lane + 64 * it;
        if (G::on(it, j)) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[it * R + r] = src(j + r * G::L);
        }
    }
}

// twiddles (exactly stockham_pass of hhsr_fft.hip) + the register butterfly
template <int N, int R, int Ns, bool PW>
__device__ __forceinline__ void w_bfly(const float2* __restrict__ twp, int lane, float2* v) {
    using G = WGeo<N, R>;
#pragma unroll
    for (int it = 0; it < G::RO; ++it) {
        const int j = lane + 64 * it;
        if (G::on(it, j)) {
            float2* x = v + it * R;
            if (Ns > 1) {
                const int k = (int)((unsigned)j % (unsigned)Ns);
                if (R <= HHSR_FFT_POW_RMAX && PW) {
                    float2 w[R];
                    w[1] = twp[k];
#pragma unroll
                    for (int r = 2; r < R; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
#pragma unroll
                    for (int r = 1; r < R; ++r) x[r] = cmul(x[r], w[r]);
                } else {
#pragma unroll
                    for (int r = 1; r < R; ++r) x[r] = cmul(x[r], twp[(r - 1) * Ns + k]);
                }
            }
            dft_reg<R>(x);
        }
    }
}

// one component (C = 0: real, 1: imaginary) of a pass's outputs to their autosorted places / of a pass's inputs back
template <int N, int R, int Ns, int C>
__device__ __forceinline__ void w_put(float* __restrict__ buf, int lane, const float2* v) {
    using G = WGeo<N, R>;
#pragma unroll
    for (int it = 0; it < G::RO; ++it) {
        const int j = lane + 64 * it;
        if (G::on(it, j)) {
            const int q = (int)((unsigned)j / (unsigned)Ns), k = j - q * Ns;
            const int d = q * (R * Ns) + k;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[d + r * Ns] = C ? v[it * R + r].y : v[it * R + r].x;
        }
    }
}
template <int N, int R, int C>
__device__ __forceinline__ void w_get(const float* __restrict__ buf, int lane, float2* v) {
    using G = WGeo<N, R>;
#pragma unroll
    for (int it = 0; it < G::RO; ++it) {
        const int j = lane + 64 * it;
        if (G::on(it, j)) {
#pragma unroll
            for (int r = 0; r < R; ++r) (C ? v[it * R + r].y : v[it * R + r].x) = buf[j + r * G::L];
        }
    }
}

// The outputs of a radix-RA pass (sub-transform length NsA before it) become the inputs of a radix-RB pass: real parts
// out, real parts in, imaginary parts out, imaginary parts in — in place in v (the imaginary parts stay in their old
// slots until they are written; the two layouts only have to fit the array).
template <int N, int RA, int NsA, int RB>
__device__ __forceinline__ void w_exchange(float* __restrict__ buf, int lane, float2* v) {
    lane = opaque(lane);
    w_put<N, RA, NsA, 0>(buf, lane, v);
    wave_fence();
    w_get<N, RB, 0>(buf, lane, v);
    wave_fence();
    w_put<N, RA, NsA, 1>(buf, lane, v);
    wave_fence();
    w_get<N, RB, 1>(buf, lane, v);
    wave_fence();
}

// Passes PASS .. NP-1 of plan P: v holds the inputs of pass PASS and ends up with the outputs of the last pass (element
// n = j + r (N / R_last) of round it at v[it R_last + r]).
template <class P, int PASS = 0>
__device__ __forceinline__ void w_fft(float* __restrict__ buf, const float2* __restrict__ tw, int lane, float2* v) {
    constexpr int R = P::R[PASS], Ns = P::ns(PASS);
    w_bfly<P::N, R, Ns, P::pw(PASS)>(tw + P::toff(PASS), opaque(lane), v);
    if constexpr (PASS + 1 < P::NP) {
        w_exchange<P::N, R, Ns, P::R[PASS + 1]>(buf, lane, v);
        w_fft<P, PASS + 1>(buf, tw, lane, v);
    }
}

// cooperative copy of a table into LDS (whole workgroup; the caller synchronises)
template <int NT>
__device__ __forceinline__ void copy_table(float2* __restrict__ dst, const float2* __restrict__ src, int n, int tid) {
    for (int i = tid; i < n; i += NT) dst[i] = src[i];
}

struct WFrames {
    const float* src[HHSR_MAX_BATCH];
    float* dst[HHSR_MAX_BATCH];
    int n;
    size_t tstride;  // float2 elements between the spectra of consecutive frames
};

// ---- rows, forward ---------------------------------------------------------------------------------------------------
// Workgroup = NW waves = NW consecutive rows per iteration, persistent (row group g = blockIdx.x + i gridDim.x).  NW = 8:
// the 16-byte pieces 8 consecutive rows write into one 128-byte line of the pair-blocked spectrum come from ONE workgroup;
// SYNC: a workgroup barrier in front of the stores, so that they also come at one time and leave the L2 as whole lines
// (free-running: 1.7 x the algorithmic write traffic).  LDS: tw[P::twlen (+1)] | twW[2 nblk] | NW x float[M]
template <class P, int NW, bool SYNC>
__global__ void __launch_bounds__(NW * 64) k_wrows_fwd(WFrames fr, int H, float2* __restrict__ Tall, int Wk,
                                                         const float2* __restrict__ twM, const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    constexpr int M = P::N, R0 = P::R[0], RL = P::R[P::NP - 1], NsL = M / RL, TWP = (P::twlen + 1) & ~1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (Wk + 1) / 2;  // pairs of kept bins (Wk <= M / 2 + 1: the host checks)
    constexpr int NBF = ((M / 2 + 2) / 2 + 63) / 64;
    float2* tw = fl;
    float2* tww = fl + TWP;  // exp(-2 pi i k / W), k < 2 nblk
    float* buf = reinterpret_cast<float*>(tww + 2 * nblk) + wave * M;
    copy_table<NW * 64>(tw, twM, P::twlen, tid);
    copy_table<NW * 64>(tww, twW, 2 * nblk, tid);
    __syncthreads();
    const int total = H * fr.n, stride = (int)gridDim.x * NW;
    const int iters = (total - (int)blockIdx.x * NW + stride - 1) / stride;  // the same for every wave of the workgroup
    for (int i = 0; i < iters; ++i) {
        const int vrow = ((int)blockIdx.x + i * (int)gridDim.x) * NW + wave;
        const bool live = vrow < total;
        const int frame = live ? vrow / H : 0, y = live ? vrow - frame * H : 0;
        float4 o[NBF];
        if (live) {
            float2 v[P::maxvals()];
            const float2* __restrict__ s = reinterpret_cast<const float2*>(fr.src[frame] + (size_t)y * (2 * M));
            w_load<M, R0>([&](int n) { return s[n]; }, opaque(lane), v);  // z[n] = x[2n] + i x[2n+1]
            w_fft<P>(buf, tw, lane, v);
            // X[k] = 1/2 [(Z[k] + conj Z[M-k]) - i w_k (Z[k] - conj Z[M-k])]: one lane per PAIR of kept bins (k_rows_fwd's
            // TB = 2 path); Z[k], Z[k+1], Z[M-k], Z[M-k-1] come back through the buffer, real parts then imaginary parts
            const int lp = opaque(lane);
            float2 z0[NBF], z1[NBF], m0[NBF], m1[NBF];
            w_put<M, RL, NsL, 0>(buf, lp, v);
            wave_fence();
#pragma unroll
            for (int it = 0; it < NBF; ++it) {
                const int b = lp + 64 * it, k = 2 * b;
                if (b < nblk) {
                    const float2 a = *reinterpret_cast<const float2*>(buf + k);
                    z0[it].x = a.x, z1[it].x = a.y, m0[it].x = buf[k == 0 ? 0 : M - k], m1[it].x = buf[M - k - 1];
                }
            }
            wave_fence();
            w_put<M, RL, NsL, 1>(buf, lp, v);
            wave_fence();
#pragma unroll
            for (int it = 0; it < NBF; ++it) {
                const int b = lp + 64 * it, k = 2 * b;
                if (b < nblk) {
                    const float2 a = *reinterpret_cast<const float2*>(buf + k);
                    z0[it].y = a.x, z1[it].y = a.y, m0[it].y = buf[k == 0 ? 0 : M - k], m1[it].y = buf[M - k - 1];
                    const float4 ww = *reinterpret_cast<const float4*>(tww + k);
                    const float2 zm0 = cconj(m0[it]), zm1 = cconj(m1[it]);
                    const float2 s0 = cadd(z0[it], zm0), d0 = mul_mi(cmul(make_float2(ww.x, ww.y), csub(z0[it], zm0)));
                    const float2 s1 = cadd(z1[it], zm1), d1 = mul_mi(cmul(make_float2(ww.z, ww.w), csub(z1[it], zm1)));
                    const float2 o0 = cscale(cadd(s0, d0), 0.5f), o1 = cscale(cadd(s1, d1), 0.5f);
                    o[it] = make_float4(o0.x, o0.y, o1.x, o1.y);
                }
            }
            wave_fence();  // (the next row's exchange writes the buffer these reads came from)
        }
        if (SYNC) __syncthreads();
        if (live) {
            float2* __restrict__ T = Tall + (size_t)frame * fr.tstride + (size_t)y * 2;
            const int lp = opaque(lane);
#pragma unroll
            for (int it = 0; it < NBF; ++it) {
                const int b = lp + 64 * it;
                if (b < nblk) *reinterpret_cast<float4*>(T + (size_t)b * H * 2) = o[it];
            }
        }
    }
}

// ---- columns: forward -> Hermitian mask + normalisation -> inverse -------------------------------------------------------
// One wave per kept column kx.  An odd Wk has a padding partner column (tstride covers it): transformed like the others,
// never read.  LDS: tw[P::twlen (+1)] | NW x float[H]
__device__ __forceinline__ bool w_kept(int u, int n) {  // (= fft_kept of hhsr_fft.hip)
    int i = u + n / 2;
    if (i >= n) i -= n;
    return i >= n / 4 && i < n - (n + 3) / 4;
}

template <class P, int NW>
__global__ void __launch_bounds__(NW * 64) k_wcols(float2* __restrict__ Tall, size_t tstride, int n_frames, int W, int Wk,
                                                     const float2* __restrict__ twH, float norm) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    constexpr int H = P::N, R0 = P::R[0], RL = P::R[P::NP - 1], TWP = (P::twlen + 1) & ~1;
    constexpr int NsL = H / RL;  // the last pass leaves element j + r NsL
    using GL = WGeo<H, RL>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* tw = fl;
    float* buf = reinterpret_cast<float*>(fl + TWP) + wave * H;
    copy_table<NW * 64>(tw, twH, P::twlen, tid);
    __syncthreads();
    const int Wc = (Wk + 1) & ~1;
    const int total = Wc * n_frames, stride = (int)gridDim.x * NW;
    const int hh = H / 2, lo = H / 4, hi = H - (H + 3) / 4;
    auto kept_y = [&](int u) {
        int i = u + hh;
        if (i >= H) i -= H;
        return i >= lo && i < hi;
    };
    for (int vc = (int)blockIdx.x * NW + wave; vc < total; vc += stride) {
        const int frame = vc / Wc, kx = vc - frame * Wc;
        float2* __restrict__ col = Tall + (size_t)frame * tstride + ((size_t)(kx / 2) * H) * 2 + (kx & 1);  // row y at col[2 y]
        float2 v[P::maxvals()];
        w_load<H, R0>([&](int n) { return col[2 * n]; }, opaque(lane), v);
        w_fft<P>(buf, tw, lane, v);
        // masked, normalised and conjugated (the inverse is conj(FFT(conj(.)))): k_cols' expression on the registers the
        // last pass left — element ky = j + r NsL
        {
            const int nx = kx == 0 ? 0 : W - kx;
            const bool fx = kx < Wk && w_kept(kx, W), fnx = kx < Wk && w_kept(nx, W);
            const int l = opaque(lane);
#pragma unroll
            for (int it = 0; it < GL::RO; ++it) {
                const int j = l + 64 * it;
                if (GL::on(it, j)) {
#pragma unroll
                    for (int r = 0; r < RL; ++r) {
                        const int ky = j + r * NsL, nky = ky == 0 ? 0 : H - ky;
                        const int m = (int)(fx && kept_y(ky)) + (int)(fnx && kept_y(nky));
                        v[it * RL + r] = cconj(cscale(v[it * RL + r], 0.5f * (float)m * norm));
                    }
                }
            }
        }
        w_exchange<H, RL, NsL, R0>(buf, lane, v);  // natural order -> the inverse's first pass
        w_fft<P>(buf, tw, lane, v);
        const int ls = opaque(lane);
#pragma unroll
        for (int it = 0; it < GL::RO; ++it) {
            const int j = ls + 64 * it;
            if (GL::on(it, j)) {
#pragma unroll
                for (int r = 0; r < RL; ++r) col[2 * (j + r * NsL)] = cconj(v[it * RL + r]);
            }
        }
    }
}

// ---- rows, inverse ---------------------------------------------------------------------------------------------------
// LDS: tw[P::twlen (+1)] | twW[M] | NW x float[M]
template <class P, int NW>
__global__ void __launch_bounds__(NW * 64) k_wrows_inv(const float2* __restrict__ Tall, int H, int Wk, WFrames fr,
                                                         const float2* __restrict__ twM, const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    constexpr int M = P::N, R0 = P::R[0], RL = P::R[P::NP - 1], NsL = M / RL, TWP = (P::twlen + 1) & ~1;
    using GL = WGeo<M, RL>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* tw = fl;
    float2* tww = fl + TWP;  // exp(-2 pi i k / W), k < M
    float* buf = reinterpret_cast<float*>(tww + M) + wave * M;
    copy_table<NW * 64>(tw, twM, P::twlen, tid);
    copy_table<NW * 64>(tww, twW, M, tid);
    __syncthreads();
    const int total = H * fr.n, stride = (int)gridDim.x * NW;
    constexpr int half = M / 2 + 1, nblk = (half + 1) / 2, NB = (nblk + 63) / 64;
    auto z_of = [&](float2 xk, float2 xm_, int k) {  // (xm_ = X[M-k], not yet conjugated) — k_rows_inv's expression
        const float2 xm = cconj(xm_);
        const float2 s = cadd(xk, xm), d = mul_pi(cmul(cconj(tww[k]), csub(xk, xm)));
        return cconj(cscale(cadd(s, d), 0.5f));
    };
    for (int vrow = (int)blockIdx.x * NW + wave; vrow < total; vrow += stride) {
        const int frame = vrow / H, y = vrow - frame * H;
        const float2* __restrict__ T = Tall + (size_t)frame * fr.tstride;
        float2* __restrict__ d2 = reinterpret_cast<float2*>(fr.dst[frame] + (size_t)y * (2 * M));
        float2 v[P::maxvals()];
        {
            // Z[k] = 1/2 [(X[k] + conj X[M-k]) + i conj(w_k) (X[k] - conj X[M-k])] with X = 0 above the kept band, stored
            // conjugated; one lane builds the pairs Z[k], Z[M-k] of two adjacent bins (k_rows_inv's TB = 2 path) and hands
            // them to the first pass through the buffer, real parts then imaginary parts
            const int lp = opaque(lane);
            float4 x4[NB];
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                const int b = lp + 64 * it;
                x4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (b < nblk && 2 * b < Wk) x4[it] = *reinterpret_cast<const float4*>(T + ((size_t)b * H + y) * 2);
            }
            float2 zk0[NB], zk1[NB], zm0[NB], zm1[NB];
#pragma unroll
            for (int it = 0; it < NB; ++it) {
                const int b = lp + 64 * it;
                if (b < nblk) {
                    const int k0 = 2 * b;
                    const float2 xa0 = make_float2(x4[it].x, x4[it].y);
                    const float2 xa1 = k0 + 1 < Wk ? make_float2(x4[it].z, x4[it].w) : make_float2(0.f, 0.f);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int k = k0 + e, mk = M - k;
                        const float2 xa = e ? xa1 : xa0;
                        float2 xb = make_float2(0.f, 0.f);
                        if (mk == k) xb = xa;
                        else if (mk < Wk) xb = T[((size_t)(mk / 2) * H + y) * 2 + (mk & 1)];
                        (e ? zk1[it] : zk0[it]) = z_of(xa, xb, k < M ? k : 0);
                        (e ? zm1[it] : zm0[it]) = (k < half && mk != k && mk < M) ? z_of(xb, xa, mk) : make_float2(0.f, 0.f);
                    }
                }
            }
            auto put = [&](auto comp) {
#pragma unroll
                for (int it = 0; it < NB; ++it) {
                    const int b = lp + 64 * it;
                    if (b < nblk) {
                        const int k0 = 2 * b;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int k = k0 + e, mk = M - k;
                            if (k < half && mk != k && mk < M) buf[mk] = comp(e ? zm1[it] : zm0[it]);
                        }
                        if (k0 + 1 < half) *reinterpret_cast<float2*>(buf + k0) = make_float2(comp(zk0[it]), comp(zk1[it]));
                        else if (k0 < half) buf[k0] = comp(zk0[it]);
                    }
                }
            };
            put([](float2 z) { return z.x; });
            wave_fence();
            w_get<M, R0, 0>(buf, lp, v);
            wave_fence();
            put([](float2 z) { return z.y; });
            wave_fence();
            w_get<M, R0, 1>(buf, lp, v);
            wave_fence();
        }
        w_fft<P>(buf, tw, lane, v);
        // x[2n] = Re z[n], x[2n+1] = -Im of the conjugate-stored z[n]: element n = j + r NsL straight from the registers
        const int ls = opaque(lane);
#pragma unroll
        for (int it = 0; it < GL::RO; ++it) {
            const int j = ls + 64 * it;
            if (GL::on(it, j)) {
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const float2 z = v[it * RL + r];
                    d2[j + r * NsL] = make_float2(z.x, -z.y);
                }
            }
        }
    }
}

// ---- the plans built in ---------------------------------------------------------------------------------------------------
// Rows: a schedule picked for ONE WAVE per transform — butterflies per pass close to a multiple of 64 (2000 = 5 5 16 5:
// 400 / 400 / 125 / 400 butterflies = 7 / 7 / 2 / 7 rounds, 89 - 98 % of the lanes busy; the workgroup kernels' 10 10 10 2
// would leave a fourth round with 8 of 64 lanes busy in every radix-10 pass: +22 % instructions), the radix-16 pass where
// its output stride (Ns = 25) no longer collides on the LDS banks.  Own twiddle tables (w_tables()).
// Columns: the workgroup kernels' schedule and tables (3000 = 3 10 10 10: 16 / 5 / 5 / 5 rounds, 94 - 98 %).
using PlanM2000 = WPlan<2000, 5, 5, 16, 5>;    // rows of a 4000-pixel-wide image
using PlanH3000 = WPlan<3000, 3, 10, 10, 10>;  // columns of a 3000-pixel-high image

template <class P>
bool plan_matches(int n, const HhsrRadices& rad) {
    if (n != P::N || rad.n != P::NP || rad.pow_min != HHSR_FFT_POW_MIN) return false;
    for (int i = 0; i < P::NP; ++i)
        if (rad.r[i] != P::R[i]) return false;
    return true;
}

// per-pass twiddle tables of plan P, concatenated (the layout pass_twiddles() of hhsr_fft.hip builds for ITS schedule)
template <class P>
float2* w_tables() {
    std::vector<float2> h;
    int Ns = 1;
    for (int p = 0; p < P::NP; ++p) {
        const int R = P::R[p];
        for (int r = 1; r < (P::pw(p) ? 2 : R); ++r)
            for (int k = 0; k < Ns; ++k) {
                const double a = -2.0 * M_PI * (double)r * (double)k / ((double)Ns * (double)R);
                h.push_back(make_float2((float)cos(a), (float)sin(a)));
            }
        Ns *= R;
    }
    float2* d = nullptr;
    if ((int)h.size() != P::twlen || hipMalloc((void**)&d, sizeof(float2) * h.size()) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    return d;
}

constexpr int ROWS_FWD_NW = 8, ROWS_INV_NW = 16, COLS_NW = 10;
constexpr bool ROWS_FWD_SYNC = true;

template <class P>
size_t rows_fwd_lds(int Wk) { return sizeof(float2) * (((P::twlen + 1) & ~1) + 2 * ((Wk + 1) / 2)) + sizeof(float) * (size_t)ROWS_FWD_NW * P::N; }
template <class P>
size_t rows_inv_lds() { return sizeof(float2) * (((P::twlen + 1) & ~1) + P::N) + sizeof(float) * (size_t)ROWS_INV_NW * P::N; }
template <class P>
size_t cols_lds() { return sizeof(float2) * ((P::twlen + 1) & ~1) + sizeof(float) * (size_t)COLS_NW * P::N; }

}  // namespace

// Which phases have a wave-synchronous kernel for this plan: bit 0 rows forward, bit 1 columns, bit 2 rows inverse.
// Called once per plan from hhsr_fft_create (allocates the row kernels' twiddle table, sets the kernels' LDS sizes).
void hhsr_fftw_prepare(HhsrFft& f) {
    f.wave = 0;
    f.twMw = nullptr;
    const int M = f.W / 2;
    if (M == PlanM2000::N && f.Wk <= M / 2 + 1 && f.radM.pow_min == HHSR_FFT_POW_MIN) {
        const size_t lf = rows_fwd_lds<PlanM2000>(f.Wk), li = rows_inv_lds<PlanM2000>();
        if (lf <= 160 * 1024 && li <= 160 * 1024 &&
            hipFuncSetAttribute((const void*)k_wrows_fwd<PlanM2000, ROWS_FWD_NW, ROWS_FWD_SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf) == hipSuccess &&
            hipFuncSetAttribute((const void*)k_wrows_inv<PlanM2000, ROWS_INV_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)li) == hipSuccess &&
            (f.twMw = w_tables<PlanM2000>()) != nullptr)
            f.wave |= 1 | 4;
    }
    if (plan_matches<PlanH3000>(f.H, f.radH)) {
        const size_t lc = cols_lds<PlanH3000>();
        if (lc <= 160 * 1024 &&
            hipFuncSetAttribute((const void*)k_wcols<PlanH3000, COLS_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lc) == hipSuccess)
            f.wave |= 2;
    }
    (void)hipGetLastError();
}

static WFrames wframes(const HhsrFft& f, const float* const* srcs, float* const* dsts, int n) {
    WFrames fr;
    fr.n = n;
    fr.tstride = f.tstride;
    for (int k = 0; k < HHSR_MAX_BATCH; ++k) {
        fr.src[k] = srcs[k < n ? k : 0];
        fr.dst[k] = dsts[k < n ? k : 0];
    }
    return fr;
}

static int cu_count() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        return cus;
    }();
    return n;
}

static int resident_grid(int groups, size_t lds) {  // one resident round of persistent workgroups
    const int per_cu = lds > 0 ? (int)((160 * 1024) / lds) : 1;
    const int slots = cu_count() * (per_cu > 0 ? per_cu : 1);
    return groups < slots ? groups : slots;
}

void hhsr_fftw_rows_fwd(const HhsrFft& f, const float* const* srcs, float* const* dsts, int n, hipStream_t s) {
    const size_t lds = rows_fwd_lds<PlanM2000>(f.Wk);
    hipLaunchKernelGGL((k_wrows_fwd<PlanM2000, ROWS_FWD_NW, ROWS_FWD_SYNC>), dim3(resident_grid(hhsr_cdiv(f.H * n, ROWS_FWD_NW), lds)),
                       dim3(ROWS_FWD_NW * 64), lds, s, wframes(f, srcs, dsts, n), f.H, f.T, f.Wk, f.twMw, f.twW);
}

void hhsr_fftw_cols(const HhsrFft& f, int n, float norm, hipStream_t s) {
    const size_t lds = cols_lds<PlanH3000>();
    hipLaunchKernelGGL((k_wcols<PlanH3000, COLS_NW>), dim3(resident_grid(hhsr_cdiv(((f.Wk + 1) & ~1) * n, COLS_NW), lds)),
                       dim3(COLS_NW * 64), lds, s, f.T, f.tstride, n, f.W, f.Wk, f.twH, norm);
}

void hhsr_fftw_rows_inv(const HhsrFft& f, const float* const* srcs, float* const* dsts, int n, hipStream_t s) {
    const size_t lds = rows_inv_lds<PlanM2000>();
    hipLaunchKernelGGL((k_wrows_inv<PlanM2000, ROWS_INV_NW>), dim3(resident_grid(hhsr_cdiv(f.H * n, ROWS_INV_NW), lds)),
                       dim3(ROWS_INV_NW * 64), lds, s, f.T, f.H, f.Wk, wframes(f, srcs, dsts, n), f.twMw, f.twW);
}

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
//   * occupancy is set by LDS (one 16-24 KB buffer per wave), which leaves every wave 256 VGPRs: the row kernels load the
//     NEXT row's first-pass inputs into registers before they run the passes of the current one.
// The butterflies, the twiddle tables and the order of every floating-point operation are those of hhsr_fft.hip
// (hhsr_fft_bfly.h; the static plans below equal the schedules its host code picks for these lengths): the results are
// bit-identical, which tests/test_hip_parity.py::test_grey_wave_kernels_equal_workgroup_kernels asserts.
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
// function of the lane alone, i.e. invariant over the kernel's row loop — hoisted out of it they are ~70 values that stay
// live across the whole loop body, get spilled, and their scratch reloads share vmcnt with the prefetched row.  Recomputing
// them per pass is a handful of instructions.
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

// Butterfly j = lane + 64 it of a radix-R pass over N points (L = N / R butterflies, RO rounds of 64): its inputs are
// x[j + r L], its outputs go to x[(j / Ns) R Ns + j % Ns + r Ns] (Stockham autosort, in place: every input of the pass is
// in registers before the first output is written).
template <int N, int R>
struct WGeo {
    static constexpr int L = N / R, RO = (L + 63) / 64;
    static __device__ __forceinline__ bool on(int it, int j) { return (it + 1) * 64 <= L || j < L; }
};

// rounds [it0, it1) of a pass: inputs into registers
template <int N, int R, typename Src>
__device__ __forceinline__ void w_read(Src src, int lane, float2* v, int it0 = 0, int it1 = WGeo<N, R>::RO) {
    using G = WGeo<N, R>;
#pragma unroll
    for (int it = 0; it < G::RO; ++it) {
        const int j = lane + 64 * it;
        if (it >= it0 && it < it1 && G::on(it, j)) {
#pragma unroll
            for (int r = 0; r < R; ++r) v[it * R + r] = src(j + r * G::L);
        }
    }
}

// twiddles (exactly stockham_pass of hhsr_fft.hip) + the register butterfly, rounds [it0, it1)
template <int N, int R, int Ns, bool PW>
__device__ __forceinline__ void w_bfly(const float2* __restrict__ twp, int lane, float2* v, int it0 = 0,
                                       int it1 = WGeo<N, R>::RO) {
    using G = WGeo<N, R>;
#pragma unroll
    for (int it = 0; it < G::RO; ++it) {
        const int j = lane + 64 * it;
        if (it >= it0 && it < it1 && G::on(it, j)) {
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

template <int N, int R, int Ns, typename Dst>
__device__ __forceinline__ void w_write(Dst dst, int lane, const float2* v) {
    using G = WGeo<N, R>;
#pragma unroll
    for (int it = 0; it < G::RO; ++it) {
        const int j = lane + 64 * it;
        if (G::on(it, j)) {
            const int q = (int)((unsigned)j / (unsigned)Ns), k = j - q * Ns;
            const int d = q * (R * Ns) + k;
#pragma unroll
            for (int r = 0; r < R; ++r) dst(d + r * Ns, v[it * R + r]);
        }
    }
}

struct LdsSrc {
    const float2* b;
    __device__ __forceinline__ float2 operator()(int i) const { return b[i]; }
};
struct LdsDst {
    float2* b;
    __device__ __forceinline__ void operator()(int i, float2 x) const { b[i] = x; }
};

// Passes P0 .. P::NP-1 of plan P over the wave's LDS buffer.  FROM_REGS: pass P0's inputs are already in v (the caller
// loaded them from global memory).  TO_REGS: the outputs of the last pass stay in v (element n = j + r (N / R) of round it
// at v[it R + r]) instead of going back to LDS.
// Rounds per group of a pass: the rounds of a group have their LDS reads (data and twiddles) in flight together; a
// scheduling barrier between groups keeps the compiler from hoisting EVERY read of the pass to its top (all inputs plus
// all twiddles of a radix-10 pass live at once: 150 VGPRs on top of the prefetched row — it spilled).
__host__ __device__ constexpr int w_group(int R) { return R >= 8 ? 1 : R >= 4 ? 3 : 8; }

template <class P, int PASS, bool FROM_REGS, bool TO_REGS>
__device__ __forceinline__ void w_passes(float2* __restrict__ buf, const float2* __restrict__ tw, int lane, float2* v) {
    if constexpr (PASS < P::NP) {
        lane = opaque(lane);
        constexpr int R = P::R[PASS], Ns = P::ns(PASS), RO = WGeo<P::N, R>::RO, GR = w_group(R);
        constexpr bool last = PASS == P::NP - 1;
#pragma unroll
        for (int g = 0; g < RO; g += GR) {
            if (!(FROM_REGS && PASS == 0)) w_read<P::N, R>(LdsSrc{buf}, lane, v, g, g + GR);
            w_bfly<P::N, R, Ns, P::pw(PASS)>(tw + P::toff(PASS), lane, v, g, g + GR);
            if (g + GR < RO && !(FROM_REGS && PASS == 0)) __builtin_amdgcn_sched_barrier(0);
        }
        if (!(last && TO_REGS)) {
            wave_fence();  // every input of the pass is in registers
            w_write<P::N, R, Ns>(LdsDst{buf}, lane, v);
            wave_fence();
        }
        w_passes<P, PASS + 1, FROM_REGS, TO_REGS>(buf, tw, lane, v);
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

// All three kernels are PERSISTENT and software-pipelined so that the one full wait of an iteration (s_waitcnt vmcnt(0)
// where the next transform's inputs are first used; gfx950 counts loads and stores in one counter and hipcc waits for
// zero at a loop head) only ever sees memory operations issued a whole transform earlier: the global stores of transform
// i - 1 are issued AFTER the first pass of transform i has consumed its prefetched inputs, the loads of transform i + 1
// right behind them.  (First version: stores at the end of the iteration, the wait at its top — 39 % of the wave time in
// s_waitcnt, profiles/r06_fft_wave_pmc.txt.)

// ---- rows, forward ---------------------------------------------------------------------------------------------------
// Workgroup = NW waves = NW CONSECUTIVE rows (the 16-byte pieces NW = 8 rows write into one 128-byte line of the
// pair-blocked spectrum leave one CU); persistent: row group g = blockIdx.x + i gridDim.x.
// LDS: tw[P::twlen (+1)] | twW[2 nblk] | NW x row[M]
template <class P, int NW>
__global__ void __launch_bounds__(NW * 64) k_wrows_fwd(WFrames fr, int H, float2* __restrict__ Tall, int Wk,
                                                         const float2* __restrict__ twM, const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    constexpr int M = P::N, R0 = P::R[0], TWP = (P::twlen + 1) & ~1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = (Wk + 1) / 2;  // pairs of kept bins (Wk <= M / 2 + 1: the host checks)
    constexpr int NBF = ((M / 2 + 2) / 2 + 63) / 64;
    float2* tw = fl;
    float2* tww = fl + TWP;                 // exp(-2 pi i k / W), k < 2 nblk
    float2* buf = tww + 2 * nblk + wave * M;
    copy_table<NW * 64>(tw, twM, P::twlen, tid);
    copy_table<NW * 64>(tww, twW, 2 * nblk, tid);
    __syncthreads();  // the only workgroup barrier of the kernel
    const int total = H * fr.n, stride = (int)gridDim.x * NW;
    int vrow = (int)blockIdx.x * NW + wave;
    float2 v[WGeo<M, R0>::RO * R0], u[P::maxvals()];
    float4 d[NBF];            // the finished bins of the previous row, stored one iteration late
    float2* dT = nullptr;     // ... to dT + 2 (b H)  (= row y of its frame's spectrum)
    auto gsrc = [&](int row) {
        const int frame = row / H, y = row - frame * H;
        return reinterpret_cast<const float2*>(fr.src[frame] + (size_t)y * (2 * M));  // z[n] = x[2n] + i x[2n+1]
    };
    auto flush = [&]() {
        const int lp = opaque(lane);
#pragma unroll
        for (int it = 0; it < NBF; ++it) {
            const int b = lp + 64 * it;
            if (b < nblk) *reinterpret_cast<float4*>(dT + (size_t)b * H * 2) = d[it];
        }
    };
    if (vrow < total) {
        const float2* __restrict__ s = gsrc(vrow);
        w_read<M, R0>([&](int i) { return s[i]; }, lane, v);
    }
    while (vrow < total) {
        const int frame = vrow / H, y = vrow - frame * H;
        // pass 0 on the registers the previous iteration (or the prologue) loaded
        w_bfly<M, R0, 1, false>(tw, lane, v);
        w_write<M, R0, 1>(LdsDst{buf}, opaque(lane), v);
        wave_fence();
        if (dT) flush();  // the previous row's bins
        const int next = vrow + stride;
        if (next < total) {  // the next row's first-pass inputs: in flight across the passes below
            const float2* __restrict__ s = gsrc(next);
            w_read<M, R0>([&](int i) { return s[i]; }, opaque(lane), v);
        }
        w_passes<P, 1, false, false>(buf, tw, lane, u);
        // X[k] = 1/2 [(Z[k] + conj Z[M-k]) - i w_k (Z[k] - conj Z[M-k])]: one lane per PAIR of kept bins (k_rows_fwd's
        // TB = 2 path, operation for operation), 16 bytes of the pair-blocked spectrum
        const int lp = opaque(lane);
#pragma unroll
        for (int it = 0; it < NBF; ++it) {
            const int b = lp + 64 * it;
            if (b < nblk) {
                const int k = 2 * b;
                const float4 zz = *reinterpret_cast<const float4*>(buf + k);
                const float2 zm0 = cconj(buf[k == 0 ? 0 : M - k]), zm1 = cconj(buf[M - k - 1]);
                const float4 ww = *reinterpret_cast<const float4*>(tww + k);
                const float2 z0 = make_float2(zz.x, zz.y), z1 = make_float2(zz.z, zz.w);
                const float2 s0 = cadd(z0, zm0), d0 = mul_mi(cmul(make_float2(ww.x, ww.y), csub(z0, zm0)));
                const float2 s1 = cadd(z1, zm1), d1 = mul_mi(cmul(make_float2(ww.z, ww.w), csub(z1, zm1)));
                const float2 o0 = cscale(cadd(s0, d0), 0.5f), o1 = cscale(cadd(s1, d1), 0.5f);
                d[it] = make_float4(o0.x, o0.y, o1.x, o1.y);
            }
        }
        dT = Tall + (size_t)frame * fr.tstride + (size_t)y * 2;
        wave_fence();  // the bins are read before the next row's pass 0 overwrites the buffer
        vrow = next;
    }
    if (dT) flush();
}

// ---- columns: forward -> Hermitian mask + normalisation -> inverse -------------------------------------------------------
// One wave per kept column kx (the two halves of the 16-byte pieces of a column pair are read and written by waves of one
// CU most of the time).  An odd Wk has a padding partner column (tstride covers it): transformed like the others, never read.
// LDS: tw[P::twlen (+1)] | NW x col[H]
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
    float2* buf = fl + TWP + wave * H;
    copy_table<NW * 64>(tw, twH, P::twlen, tid);
    __syncthreads();  // the only workgroup barrier of the kernel
    const int Wc = (Wk + 1) & ~1;
    const int total = Wc * n_frames, stride = (int)gridDim.x * NW;
    const int hh = H / 2, lo = H / 4, hi = H - (H + 3) / 4;
    auto kept_y = [&](int u) {
        int i = u + hh;
        if (i >= H) i -= H;
        return i >= lo && i < hi;
    };
    auto colptr = [&](int vc) {
        const int frame = vc / Wc, kx = vc - frame * Wc;
        return Tall + (size_t)frame * tstride + ((size_t)(kx / 2) * H) * 2 + (kx & 1);  // row y at [2 y]
    };
    float2 v[WGeo<H, R0>::RO * R0], u[P::maxvals()];
    float2* __restrict__ dcol = nullptr;  // the previous column: its result is stored one iteration late
    auto flush = [&]() {
        const int ls = opaque(lane);
#pragma unroll
        for (int it = 0; it < GL::RO; ++it) {
            const int j = ls + 64 * it;
            if (GL::on(it, j)) {
#pragma unroll
                for (int r = 0; r < RL; ++r) dcol[2 * (j + r * NsL)] = cconj(u[it * RL + r]);
            }
        }
    };
    int vc = (int)blockIdx.x * NW + wave;
    if (vc < total) {
        const float2* __restrict__ col = colptr(vc);
        w_read<H, R0>([&](int i) { return col[2 * i]; }, lane, v);
    }
    while (vc < total) {
        const int frame = vc / Wc, kx = vc - frame * Wc;
        w_bfly<H, R0, 1, false>(tw, lane, v);
        w_write<H, R0, 1>(LdsDst{buf}, opaque(lane), v);
        wave_fence();
        if (dcol) flush();
        const int next = vc + stride;
        if (next < total) {
            const float2* __restrict__ col = colptr(next);
            w_read<H, R0>([&](int i) { return col[2 * i]; }, opaque(lane), v);
        }
        w_passes<P, 1, false, true>(buf, tw, lane, u);
        wave_fence();
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
                        buf[ky] = cconj(cscale(u[it * RL + r], 0.5f * (float)m * norm));
                    }
                }
            }
        }
        wave_fence();
        w_passes<P, 0, false, true>(buf, tw, lane, u);
        wave_fence();  // (the next column's pass 0 writes the buffer the last pass read)
        dcol = colptr(vc);
        vc = next;
    }
    if (dcol) flush();
}

// ---- rows, inverse ---------------------------------------------------------------------------------------------------
// LDS: tw[P::twlen (+1)] | twW[M] | NW x row[M]
template <class P, int NW>
__global__ void __launch_bounds__(NW * 64) k_wrows_inv(const float2* __restrict__ Tall, int H, int Wk, WFrames fr,
                                                         const float2* __restrict__ twM, const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    constexpr int M = P::N, RL = P::R[P::NP - 1], NsL = M / RL, TWP = (P::twlen + 1) & ~1;
    using GL = WGeo<M, RL>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float2* tw = fl;
    float2* tww = fl + TWP;  // exp(-2 pi i k / W), k < M
    float2* buf = tww + M + wave * M;
    copy_table<NW * 64>(tw, twM, P::twlen, tid);
    copy_table<NW * 64>(tww, twW, M, tid);
    __syncthreads();  // the only workgroup barrier of the kernel
    const int total = H * fr.n, stride = (int)gridDim.x * NW;
    constexpr int half = M / 2 + 1, nblk = (half + 1) / 2, NB = (nblk + 63) / 64;
    auto z_of = [&](float2 xk, float2 xm_, int k) {  // (xm_ = X[M-k], not yet conjugated) — k_rows_inv's expression
        const float2 xm = cconj(xm_);
        const float2 s = cadd(xk, xm), d = mul_pi(cmul(cconj(tww[k]), csub(xk, xm)));
        return cconj(cscale(cadd(s, d), 0.5f));
    };
    float4 x4[NB];
    auto load_bins = [&](int row) {  // the row's kept bins, one 16-byte pair per lane and round
        const int frame = row / H, y = row - frame * H;
        const float2* __restrict__ T = Tall + (size_t)frame * fr.tstride;
        const int l = opaque(lane);
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int b = l + 64 * it;
            x4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b < nblk && 2 * b < Wk) x4[it] = *reinterpret_cast<const float4*>(T + ((size_t)b * H + y) * 2);
        }
    };
    // Z[k] = 1/2 [(X[k] + conj X[M-k]) + i conj(w_k) (X[k] - conj X[M-k])] with X = 0 above the kept band, stored
    // conjugated; one lane builds the pairs Z[k], Z[M-k] of two adjacent bins (k_rows_inv's TB = 2 path)
    auto spectrum_to_lds = [&](int row) {
        const int frame = row / H, y = row - frame * H;
        const float2* __restrict__ T = Tall + (size_t)frame * fr.tstride;
        const int lp = opaque(lane);
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int b = lp + 64 * it;
            if (b < nblk) {
                const int k0 = 2 * b;
                const float2 xa0 = make_float2(x4[it].x, x4[it].y);
                const float2 xa1 = k0 + 1 < Wk ? make_float2(x4[it].z, x4[it].w) : make_float2(0.f, 0.f);
                float2 zk[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int k = k0 + e, mk = M - k;
                    const float2 xa = e ? xa1 : xa0;
                    float2 xb = make_float2(0.f, 0.f);
                    if (mk == k) xb = xa;
                    else if (mk < Wk) xb = T[((size_t)(mk / 2) * H + y) * 2 + (mk & 1)];
                    zk[e] = z_of(xa, xb, k < M ? k : 0);
                    if (k < half && mk != k && mk < M) buf[mk] = z_of(xb, xa, mk);
                }
                if (k0 + 1 < half) *reinterpret_cast<float4*>(buf + k0) = make_float4(zk[0].x, zk[0].y, zk[1].x, zk[1].y);
                else if (k0 < half) buf[k0] = zk[0];
            }
        }
    };
    int vrow = (int)blockIdx.x * NW + wave;
    if (vrow < total) {
        load_bins(vrow);
        spectrum_to_lds(vrow);
        wave_fence();
        if (vrow + stride < total) load_bins(vrow + stride);
    }
    float2 v[P::maxvals()];
    while (vrow < total) {
        const int frame = vrow / H, y = vrow - frame * H;
        float2* __restrict__ d2 = reinterpret_cast<float2*>(fr.dst[frame] + (size_t)y * (2 * M));
        w_passes<P, 0, false, true>(buf, tw, lane, v);
        wave_fence();  // the last pass has read the buffer: the next row's spectrum may overwrite it
        const int next = vrow + stride;
        if (next < total) spectrum_to_lds(next);  // (consumes the bins loaded one iteration ago)
        wave_fence();
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
        if (next + stride < total) load_bins(next + stride);
        vrow = next;
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

constexpr int ROWS_NW = 8, COLS_NW = 4;

template <class P>
size_t rows_fwd_lds(int Wk) { return sizeof(float2) * (((P::twlen + 1) & ~1) + 2 * ((Wk + 1) / 2) + (size_t)ROWS_NW * P::N); }
template <class P>
size_t rows_inv_lds() { return sizeof(float2) * (((P::twlen + 1) & ~1) + P::N + (size_t)ROWS_NW * P::N); }
template <class P>
size_t cols_lds() { return sizeof(float2) * (((P::twlen + 1) & ~1) + (size_t)COLS_NW * P::N); }

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
            hipFuncSetAttribute((const void*)k_wrows_fwd<PlanM2000, ROWS_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lf) == hipSuccess &&
            hipFuncSetAttribute((const void*)k_wrows_inv<PlanM2000, ROWS_NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)li) == hipSuccess &&
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

void hhsr_fftw_rows_fwd(const HhsrFft& f, const float* const* srcs, float* const* dsts, int n, hipStream_t s) {
    const int groups = hhsr_cdiv(f.H * n, ROWS_NW), grid = groups < cu_count() ? groups : cu_count();
    hipLaunchKernelGGL((k_wrows_fwd<PlanM2000, ROWS_NW>), dim3(grid), dim3(ROWS_NW * 64), rows_fwd_lds<PlanM2000>(f.Wk), s,
                       wframes(f, srcs, dsts, n), f.H, f.T, f.Wk, f.twMw, f.twW);
}

void hhsr_fftw_cols(const HhsrFft& f, int n, float norm, hipStream_t s) {
    const int groups = hhsr_cdiv(((f.Wk + 1) & ~1) * n, COLS_NW), grid = groups < cu_count() ? groups : cu_count();
    hipLaunchKernelGGL((k_wcols<PlanH3000, COLS_NW>), dim3(grid), dim3(COLS_NW * 64), cols_lds<PlanH3000>(), s, f.T, f.tstride, n,
                       f.W, f.Wk, f.twH, norm);
}

void hhsr_fftw_rows_inv(const HhsrFft& f, const float* const* srcs, float* const* dsts, int n, hipStream_t s) {
    const int groups = hhsr_cdiv(f.H * n, ROWS_NW), grid = groups < cu_count() ? groups : cu_count();
    hipLaunchKernelGGL((k_wrows_inv<PlanM2000, ROWS_NW>), dim3(grid), dim3(ROWS_NW * 64), rows_inv_lds<PlanM2000>(), s, f.T, f.H,
                       f.Wk, wframes(f, srcs, dsts, n), f.twMw, f.twW);
}

// Fused in-LDS FFT low-pass for the grey image (Alg. 3, reference utils_image.py:82-100).
//
// The library route (hhsr_grey.hip) spends 9 full passes over the image per frame (row FFT, real-FFT
// post-processing + transpose, column FFT, transpose, mask, and the same backwards).  The mask keeps only
// |kx| <= W/4, |ky| <= H/4, so this file does the round trip in THREE kernels, each keeping a whole 1-D
// transform in LDS (in-place Stockham passes through registers, radices 2..16, per-pass twiddles from an LDS table):
//   k_rows_fwd   2 rows per workgroup (simultaneously): real row -> half-length complex FFT -> real-FFT
//                post-processing; only the Wk = W/4 + 1 kept x-bins are written, blocked-transposed
//                (x-bins in blocks of 8, 64-byte runs);
//   k_cols       two kept columns per workgroup: forward FFT over y -> Hermitian mask m'(ky, kx) and the
//                1/(H W) normalisation -> inverse FFT, in place (24 MB in / out at 12 MP);
//   k_rows_inv   2 rows per workgroup: gather the kept bins, rebuild the half-length spectrum, inverse
//                FFT, write the real row.
// HBM / MALL traffic per 12 MP frame: 48 + 24 | 24 + 24 | 24 + 48 = 192 MB (library route: ~860 MB).
// Measured at 3000x4000: 146 us (44 + 58 + 43; library plans: 222 us); error vs float64 at the library's level.
// The kernels are bound by the latency of their barrier phases, not by HBM: composite radices (4 passes for 2000
// and 3000 points instead of 5 and 6) and in-place passes (half the LDS: 3 / 2 resident workgroups per CU for the
// row / column kernels instead of 2 / 1) brought them from 171 us.
// Supported when W is even and W/2 and H factor into {2, 3, 5, 7} and the LDS budgets fit; the caller falls
// back to the library plans otherwise.  Numerics: float32 butterflies, float64-computed twiddle tables.
#include "hhsr_common.h"
#include "hhsr_fft.h"
#include <math.h>
#include <type_traits>
#include <vector>

#include "hhsr_fft_bfly.h"

// Butterflies one thread may hold between the read and the write phase of a pass (2 R VGPRs each)
__host__ __device__ constexpr int fft_maxit(int R) { return R <= 3 ? 4 : R <= 5 ? 3 : R <= 12 ? 2 : 1; }

// One Stockham pass of radix R over NB independent length-N transforms stored `bstride` elements apart, IN PLACE:
// every thread reads the inputs of its (up to MAXIT) butterflies into registers, the workgroup synchronises, and
// the autosorted outputs overwrite the same buffer (a second barrier ends the pass).  Half the LDS of the usual
// ping-pong — twice the resident workgroups for kernels that are bound by barrier / memory latency.
// twp: this pass's twiddles, twp[(r-1)*Ns + k] = exp(-2 pi i r k / (Ns R)) — contiguous in k, so the lanes of
// a wave (consecutive butterflies -> consecutive k) read consecutive LDS words (no bank conflicts).
// pw: this pass's table holds only w^k = exp(-2 pi i k / (Ns R)), k < Ns — the twiddles of inputs 2 .. R-1 are its powers,
// formed in registers by a product tree (w^r = w^(r/2) w^(r - r/2): depth <= 4, ~4e-7 relative).  The LAST passes' tables
// are (R - 1) N / R entries each — as large as the data they serve: 6000-point columns took 96 kB of LDS (one workgroup
// per CU), 4000-point row pairs 96 kB; with the power form they take 58 / 74 kB (two per CU).
template <int R>
__device__ __forceinline__ void stockham_pass(float2* __restrict__ buf, int bstride, int NB,
                                              const float2* __restrict__ twp, int N, int Ns, int tid, int nt, bool pw) {
    constexpr int MAXIT = fft_maxit(R);
    const int L = N / R, total = NB * L;
    const float rNs = 1.0f / (float)Ns, rL = 1.0f / (float)L;
    float2 v[MAXIT][R];
    int dst[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int jj = tid + it * nt;
        if (jj < total) {
            // (index arithmetic in 24-bit multiplies and 32-bit offsets: v_mul_lo_u32 and the 64-bit v_mad_u64_u32 of a
            // size_t offset issue at a quarter of the v_mad_u32_u24 rate, and these kernels are VALU-bound)
            const int bidx = NB > 1 ? (int)(((float)jj + 0.5f) * rL) : 0;  // exact floor for jj < 2^16 ... guarded by the host
            const int j = jj - __mul24(bidx, L);
            const int q = (int)(((float)j + 0.5f) * rNs);
            const int k = j - __mul24(q, Ns);
            const int base = __mul24(bidx, bstride);
            const float2* ib = buf + base + j;
#pragma unroll
            for (int r = 0; r < R; ++r) v[it][r] = ib[__mul24(r, L)];
            if (Ns > 1) {
                if (R <= HHSR_FFT_POW_RMAX && pw) {  // (compile time: the large radices never get the power form's registers)
                    float2 w[R];
                    w[1] = twp[k];
#pragma unroll
                    for (int r = 2; r < R; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
#pragma unroll
                    for (int r = 1; r < R; ++r) v[it][r] = cmul(v[it][r], w[r]);
                } else {
#pragma unroll
                    for (int r = 1; r < R; ++r) v[it][r] = cmul(v[it][r], twp[__mul24(r - 1, Ns) + k]);
                }
            }
            dft_reg<R>(v[it]);
            dst[it] = base + __mul24(q * R, Ns) + k;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int jj = tid + it * nt;
        if (jj < total) {
#pragma unroll
            for (int r = 0; r < R; ++r) buf[dst[it] + __mul24(r, Ns)] = v[it][r];
        }
    }
    __syncthreads();
}

// Forward FFTs, in place, of NB length-N sequences (sequence b at buf + b*bstride).  Every thread of the
// workgroup calls it; it starts and ends with a barrier.
__device__ __forceinline__ void fft_lds(float2* buf, int bstride, int NB, const float2* tw, int N,
                                        const HhsrRadices& rad, int tid, int nt) {
    int Ns = 1, toff = 0;
    __syncthreads();
    for (int p = 0; p < rad.n; ++p) {
        const int R = rad.r[p];
        // opaque copy of N: keeps the per-radix invariants (N / R, reciprocals, ...) of ALL switch arms from being
        // hoisted out of the pass loop, where they stay live together and spill
        int Np = N;
        asm volatile("" : "+s"(Np));
        const bool pw = rad.pow_min > 0 && R <= HHSR_FFT_POW_RMAX && (R - 1) * Ns > rad.pow_min;  // (= pass_twiddles on the host)
        switch (R) {
#define HHSR_PASS(RR) case RR: stockham_pass<RR>(buf, bstride, NB, tw + toff, Np, Ns, tid, nt, pw); break;
            HHSR_PASS(2) HHSR_PASS(3) HHSR_PASS(4) HHSR_PASS(5) HHSR_PASS(6) HHSR_PASS(7) HHSR_PASS(8) HHSR_PASS(9)
            HHSR_PASS(10) HHSR_PASS(12) HHSR_PASS(14) HHSR_PASS(15) HHSR_PASS(16)
#undef HHSR_PASS
            default: break;  // the host only schedules the radices above
        }
        toff += pw ? Ns : (R - 1) * Ns;
        Ns *= R;
    }
}

// ---- static plans: the same passes with N, the radices and every sub-transform length as template constants ------------
// The run-time passes above compute every LDS address (input leg r at j + r L, output at q R Ns + k + r Ns, twiddle at
// (r - 1) Ns + k: L, Ns in registers) and j / Ns by a float trick, per pass and per row; with constants the legs sit at
// IMMEDIATE offsets of one base address per pass, and those bases depend on the thread id alone — the compiler keeps
// them across the persistent row loop.  Same butterflies, same tables, same order of operations.
template <int N_, int... RS>
struct SPlan {
    static constexpr int N = N_;
    static constexpr int NP = (int)sizeof...(RS);
    static constexpr int R[NP] = {RS...};
    static constexpr int ns(int p) {  // sub-transform length before pass p
        int s = 1;
        for (int i = 0; i < p; ++i) s *= R[i];
        return s;
    }
    static constexpr bool pw(int p) {  // (= pass_twiddles() with the default pow_min)
        return HHSR_FFT_POW_MIN > 0 && R[p] <= HHSR_FFT_POW_RMAX && (R[p] - 1) * ns(p) > HHSR_FFT_POW_MIN;
    }
    static constexpr int toff(int p) {
        int o = 0;
        for (int i = 0; i < p; ++i) o += pw(i) ? ns(i) : (R[i] - 1) * ns(i);
        return o;
    }
    static bool matches(int n, const HhsrRadices& rad) {
        if (n != N || rad.n != NP || rad.pow_min != HHSR_FFT_POW_MIN) return false;
        for (int i = 0; i < NP; ++i)
            if (rad.r[i] != R[i]) return false;
        return true;
    }
};
struct NoPlan {
    static constexpr int N = 0;
};
// the transform length as the kernel sees it: the plan's constant, or the run-time argument
template <class SP>
__device__ __forceinline__ int plan_len(int runtime) {
    return std::is_same<SP, NoPlan>::value ? runtime : SP::N;
}

template <int N, int NB, int NT, int R, int Ns, bool PW>
__device__ __forceinline__ void stockham_pass_s(float2* __restrict__ buf, const float2* __restrict__ twp, int tid) {
    constexpr int MAXIT = fft_maxit(R), L = N / R, total = NB * L;
    float2 v[MAXIT][R];
    int dst[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int jj = tid + it * NT;
        if (it * NT < total && jj < total) {
            const int bidx = NB > 1 ? (int)((unsigned)jj / (unsigned)L) : 0;
            const int j = jj - bidx * L;
            const int q = (int)((unsigned)j / (unsigned)Ns), k = j - q * Ns;
            const float2* ib = buf + bidx * N + j;
#pragma unroll
            for (int r = 0; r < R; ++r) v[it][r] = ib[r * L];
            if (Ns > 1) {
                if (R <= HHSR_FFT_POW_RMAX && PW) {
                    float2 w[R];
                    w[1] = twp[k];
#pragma unroll
                    for (int r = 2; r < R; ++r) w[r] = cmul(w[r >> 1], w[r - (r >> 1)]);
#pragma unroll
                    for (int r = 1; r < R; ++r) v[it][r] = cmul(v[it][r], w[r]);
                } else {
#pragma unroll
                    for (int r = 1; r < R; ++r) v[it][r] = cmul(v[it][r], twp[(r - 1) * Ns + k]);
                }
            }
            dft_reg<R>(v[it]);
            dst[it] = bidx * N + q * (R * Ns) + k;
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int jj = tid + it * NT;
        if (it * NT < total && jj < total) {
#pragma unroll
            for (int r = 0; r < R; ++r) buf[dst[it] + r * Ns] = v[it][r];
        }
    }
    __syncthreads();
}

template <class P, int NB, int NT, int PASS = 0>
__device__ __forceinline__ void fft_lds_s(float2* buf, const float2* tw, int tid) {
    if (PASS == 0) __syncthreads();
    if constexpr (PASS < P::NP) {
        stockham_pass_s<P::N, NB, NT, P::R[PASS], P::ns(PASS), P::pw(PASS)>(buf, tw + P::toff(PASS), tid);
        fft_lds_s<P, NB, NT, PASS + 1>(buf, tw, tid);
    }
}

// the lengths with a static plan (= what factorize() picks for them: hhsr_fft_create checks)
// X(id, length, rows per workgroup / columns per workgroup, threads, radices...)
#define HHSR_STATIC_ROWS(X)                                                                 \
    X(1, 2000, 1, FFT_NT_SMALL, 10, 10, 10, 2) /* 4000-pixel rows (12 MP 4:3) */             \
    X(2, 4000, 1, FFT_NT, 10, 10, 10, 4)       /* 8000-pixel rows (48 MP) */                 \
    X(3, 2016, 1, FFT_NT_SMALL, 14, 12, 12)    /* 4032-pixel rows (the common 12 MP sensor) */
#define HHSR_STATIC_COLS(X)                                                                 \
    X(1, 3000, 2, FFT_NT, 3, 10, 10, 10)       /* 3000-pixel columns */                      \
    X(2, 6000, 1, FFT_NT, 10, 10, 10, 6)       /* 6000-pixel columns */                      \
    X(3, 3024, 2, FFT_NT, 9, 8, 7, 6)          /* 3024-pixel columns */

__device__ __forceinline__ bool fft_kept(int u, int n) {
    int i = u + n / 2;
    if (i >= n) i -= n;
    return i >= n / 4 && i < n - (n + 3) / 4;
}

// Layout of the kept half spectrum T: x-bins in blocks of TB, element (kx, y) at ((kx/TB)*H + y)*TB + kx%TB — the
// row kernels move runs of TB consecutive bins of one row (two adjacent rows of a workgroup are adjacent in memory) and a
// column pair is a walk at a stride of TB elements whose cache lines are shared by the columns of its block (placed on
// one XCD, see k_cols).
// x-bins per block.  The row kernels want long runs of consecutive bins of a row (8: 64 bytes), the column kernel wants its
// column pair contiguous over y (2).  Timed per phase with wall_clock64: at 8 the column kernel spends 6 + 14-19 us of its
// 57 us in its 16-byte accesses at a 64-byte stride (the transforms themselves: 14 us).  Measured at 12 MP, us per
// frame (rows fwd / cols / rows inv): 8: 39 / 57 / 39 = 135;  4: 40 / 43 / 44 = 127;  2: 43 / 35 / 59 = 137 — and with
// the row kernels walking T in memory order and k_rows_inv reading every bin once (pairs Z[k], Z[M-k]):
// 4: 40 / 44 / 40 = 124;  2: 42 / 35 / 44 = 121 (step 8.92 vs 8.99 ms).
constexpr int TB = 2;
__device__ __forceinline__ size_t t_index(int kx, int y, int H) { return ((size_t)(kx / TB) * H + y) * TB + (kx % TB); }

constexpr int FFT_NT = 512;          // threads per workgroup: column kernel, and the row kernels' long rows
// Row kernels, rows that fit: ONE row per 256-thread workgroup — 16 KB of data + the twiddles = 32 KB of LDS, five
// workgroups per CU instead of three 512-thread row pairs.  The phases of a workgroup (load a row / barrier-separated
// passes / store) only overlap with OTHER workgroups' phases, and five independent ones overlap better than three:
// measured at 12 MP (tools/ab.sh --kernels "k_rows|k_cols" default rows512,HHSR_FFT_NT_ROWS=512) rows fwd 96.3 -> 85.1 us, rows inv 101.5 -> 89.9 us per 3-4 frame launch
// (four resident workgroups: 91.9 / 97.2; six — more than fit — 97.1 / 104.7).  The column kernel LOSES with 256 threads
// (one column per workgroup: 90 -> 111 us) and keeps 512.
constexpr int FFT_NT_SMALL = 256;
// (Round 3 A/B, removed: pass twiddles from global memory instead of the LDS copy — four workgroups per CU, 248 / 221 us per
// 3-4 frame launch instead of 118 / 116: the twiddle loads sit on every pass's critical path.)
// row kernels: 512 threads, 48 KB of LDS -> three workgroups per CU need <= 85 VGPRs (6 waves per SIMD);
//              256 threads, 32 KB -> five workgroups per CU = 5 waves per SIMD (<= 102 VGPRs)
__host__ __device__ constexpr int fft_rows_wpe(int nt) { return nt == FFT_NT ? 6 : 5; }
constexpr int FFT_COLS_WPE = 4;     // column kernel: 72 KB of LDS -> two workgroups per CU (<= 128 VGPRs)

// n elements global -> LDS (or any load / store pair) with the loads of a 4-iteration batch all in flight before the first
// store: written as load -> store per iteration the compiler keeps ONE load outstanding, and a phase of 4-6 iterations
// costs 4-6 memory round trips (timed with wall_clock64: 4.8 us to load two rows, 7.6-9.4 us to gather a row pair's bins)
template <typename V, int NT = FFT_NT, class Load, class Store>
__device__ __forceinline__ void batched_for(int n, int tid, Load load, Store store) {
    for (int base = tid; base < n; base += 4 * NT) {
        V v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NT;
            if (i < n) v[u] = load(i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NT;
            if (i < n) store(i, v[u]);
        }
    }
}

// Row block of persistent workgroup `wg` in its round `it` (< 0: none left).  A 128-byte line of the blocked spectrum
// holds one column block of 16 / TB consecutive rows, i.e. of G = 16 / (TB RB) consecutive row blocks: they go to
// consecutive workgroups of ONE XCD (workgroup b runs on XCD b % 8 — observed; locality only), so that one L2 gathers /
// serves the line instead of G of them (k_rows_inv fetched 4.6 x the spectrum's bytes with the plain round-robin walk).
template <int RB>
__device__ __forceinline__ int row_block(int it, int nblocks) {
    constexpr int G = (16 / (TB * RB)) > 0 ? 16 / (TB * RB) : 1;
    int blk;
    if ((gridDim.x & 7) == 0) {
        const int v = (int)(blockIdx.x >> 3) + it * (int)(gridDim.x >> 3);
        blk = ((v / G) * 8 + (int)(blockIdx.x & 7)) * G + v % G;  // increasing in `it`
    } else {
        blk = (int)blockIdx.x + it * (int)gridDim.x;
    }
    return blk < nblocks ? blk : -1;
}

// idx -> (rb, n) = (idx / len, idx % len) for rb < RB without the ~30-instruction integer division (these kernels are
// VALU-bound: round 4).  rlen = 1.0f / len; exact for idx < 2^16 (hhsr_fft_create checks RB * len).
template <int RB>
__device__ __forceinline__ void split_row(int idx, int len, float rlen, int& rb, int& n) {
    if (RB == 1) {
        rb = 0;
        n = idx;
    } else if (RB == 2) {
        rb = idx >= len;
        n = rb ? idx - len : idx;
    } else {
        rb = (int)(((float)idx + 0.5f) * rlen);
        n = idx - __mul24(rb, len);
    }
}

// Row kernels: RB rows per workgroup, transformed SIMULTANEOUSLY (RB x fewer barriers, RB x more independent
// butterflies per thread).  LDS: tw[twlen] | RB x row[M]  (float2 each).
// Frames of a batch: the row kernels walk the row blocks of ALL frames of the launch (block b of frame f is virtual block
// f * nb + b), the column kernel takes the frame from blockIdx.y; frame f's spectrum is T + f * tstride.
struct FftFrames {
    const float* src[HHSR_MAX_BATCH];
    float* dst[HHSR_MAX_BATCH];
    int n;
    size_t tstride;  // float2 elements between the spectra of consecutive frames
};

template <int RB, int NT, class SP = NoPlan>
__global__ void __launch_bounds__(NT, fft_rows_wpe(NT)) k_rows_fwd(FftFrames fr, int H, int W,
                                                                        float2* __restrict__ Tall, int Wk, HhsrRadices rad,
                                                                        const float2* __restrict__ twM, int twlen,
                                                                        const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int M = plan_len<SP>(W / 2), tid = threadIdx.x;
    W = 2 * M;
    float2* tw = fl;
    float2* buf = tw + ((twlen + 1) & ~1);  // 16-byte aligned
    batched_for<float2, NT>(twlen, tid, [&](int k) { return twM[k]; }, [&](int k, float2 v) { tw[k] = v; });
    // persistent workgroups: the grid is one resident round (HHSR_FFT_PERSIST), every workgroup walks the row blocks
    // (row_block): one twiddle copy and one dispatch per workgroup slot instead of per block
    const int nb = (H + RB - 1) / RB;
    for (int it = 0, vblk; (vblk = row_block<RB>(it, nb * fr.n)) >= 0; ++it) {
    if (it) __syncthreads();  // the previous block's stores have read the buffer
    const int frame = vblk / nb, blk = vblk - frame * nb;
    const float* __restrict__ src = fr.src[frame];
    float2* __restrict__ T = Tall + (size_t)frame * fr.tstride;
    const int y0 = blk * RB;
    const int nrows = min(RB, H - y0);
    if ((M & 1) == 0 && (W & 3) == 0) {  // two complex samples (16 bytes) per lane
        const int Mh = M / 2;
        const float rMh = 1.0f / (float)Mh;
        const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src + (size_t)y0 * W);  // rows are Mh float4 apart
        batched_for<float4, NT>(nrows * Mh, tid, [&](int idx) { return src4[idx]; },
                            [&](int idx, float4 v) {
                                int rb, n;
                                split_row<RB>(idx, Mh, rMh, rb, n);
                                reinterpret_cast<float4*>(buf + __mul24(rb, M))[n] = v;
                            });
    } else {
        for (int idx = tid; idx < nrows * M; idx += NT) {  // z[n] = x[2n] + i x[2n+1]
            const int rb = idx / M, n = idx - rb * M;
            buf[rb * M + n] = reinterpret_cast<const float2*>(src + (size_t)(y0 + rb) * W)[n];
        }
    }
    if constexpr (std::is_same<SP, NoPlan>::value) fft_lds(buf, M, nrows, tw, M, rad, tid, NT);
    else fft_lds_s<SP, RB, NT>(buf, tw, tid);  // (RB = 1: nrows is always RB)
    // X[k] = 1/2 [(Z[k] + conj Z[M-k]) - i w_k (Z[k] - conj Z[M-k])],  w_k = exp(-2 pi i k / W); kept bins only,
    // straight from LDS to the blocked-transposed spectrum
    const int nblk = (Wk + TB - 1) / TB;
    if (TB == 2 && (M & 1) == 0 && nrows == RB) {
        // one thread per (bin pair, row): Z[k], Z[k+1] as one 16-byte LDS read, the pair of kept bins as one 16-byte
        // store — the rows of the block are adjacent in T, so RB consecutive lanes write RB x 16 contiguous bytes.  (An odd
        // Wk writes bin Wk into T's padding: tstride covers ceil(Wk / 8) * 8 bins and nobody reads it.)
        for (int idx = tid; idx < nblk * RB; idx += NT) {
            const int b = idx / RB, rb = idx - b * RB, k = 2 * b;  // (RB is a power of two)
            const float2* Z = buf + __mul24(rb, M);
            const float4 zz = *reinterpret_cast<const float4*>(Z + k);
            const float2 zm0 = cconj(Z[k == 0 ? 0 : M - k]), zm1 = cconj(Z[M - k - 1]);
            const float4 ww = *reinterpret_cast<const float4*>(twW + k);
            const float2 z0 = make_float2(zz.x, zz.y), z1 = make_float2(zz.z, zz.w);
            const float2 s0 = cadd(z0, zm0), d0 = mul_mi(cmul(make_float2(ww.x, ww.y), csub(z0, zm0)));
            const float2 s1 = cadd(z1, zm1), d1 = mul_mi(cmul(make_float2(ww.z, ww.w), csub(z1, zm1)));
            const float2 o0 = cscale(cadd(s0, d0), 0.5f), o1 = cscale(cadd(s1, d1), 0.5f);
            *reinterpret_cast<float4*>(T + ((size_t)b * H + y0 + rb) * 2) = make_float4(o0.x, o0.y, o1.x, o1.y);
        }
    } else
    for (int idx = tid; idx < nblk * nrows * TB; idx += NT) {  // in the order the bins lie in T (see k_rows_inv)
        const int b = idx / (nrows * TB), q = idx - b * (nrows * TB);
        const int rb = q / TB, k = b * TB + (q - rb * TB);
        if (k >= Wk) continue;
        const float2* Z = buf + rb * M;
        const float2 zk = Z[k == M ? 0 : k], zm = cconj(Z[k == 0 ? 0 : M - k]);
        const float2 s = cadd(zk, zm), d = mul_mi(cmul(twW[k], csub(zk, zm)));
        T[t_index(k, y0 + rb, H)] = cscale(cadd(s, d), 0.5f);
    }
    }
}

// Column kernel: TWO adjacent kept columns per workgroup (one 16-byte load per row serves both), transformed
// simultaneously.  LDS: tw[twlen] | 2 x col[H]
template <int NC, class SP = NoPlan>  // kept columns per workgroup: 2 (one 16-byte load per row serves both) or 1 (long columns)
__global__ void __launch_bounds__(FFT_NT, FFT_COLS_WPE) k_cols(float2* __restrict__ Tall, size_t tstride, int H, int W, int Wk,
                                                                    HhsrRadices rad, const float2* __restrict__ twH,
                                                                    int twlen, float norm) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int tid = threadIdx.x;
    H = plan_len<SP>(H);
    // workgroup b runs on XCD b % 8 (observed; locality only): give the 4 column pairs of one 64-byte block to
    // 4 consecutive workgroups of ONE XCD so that its L2 serves each cache line to all of them
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int per = 8 / NC;  // workgroups per 8-column block
    const int kx = 8 * (xcd + 8 * (loc / per)) + NC * (loc % per);
    if (kx >= Wk) return;
    float2* __restrict__ T = Tall + (size_t)blockIdx.y * tstride;  // blockIdx.y: frame of the batch
    float2* tw = fl;
    float2* buf = tw + ((twlen + 1) & ~1);  // 16-byte aligned
    float2* colb = T + ((size_t)(kx / TB) * H) * TB + (kx % TB);  // row y at colb[TB y]
    float4* col = reinterpret_cast<float4*>(colb);                 // NC = 2: row y at col[(TB / 2) y]
    batched_for<float2>(twlen, tid, [&](int k) { return twH[k]; }, [&](int k, float2 v) { tw[k] = v; });
    if (NC == 2) {
        batched_for<float4>(H, tid, [&](int k) { return col[(size_t)(TB / 2) * k]; },
                            [&](int k, float4 v) {
                                buf[k] = make_float2(v.x, v.y);
                                buf[H + k] = make_float2(v.z, v.w);
                            });
    } else {
        batched_for<float2>(H, tid, [&](int k) { return colb[(size_t)TB * k]; }, [&](int k, float2 v) { buf[k] = v; });
    }
    if constexpr (std::is_same<SP, NoPlan>::value) fft_lds(buf, H, NC, tw, H, rad, tid, FFT_NT);
    else fft_lds_s<SP, NC, FFT_NT>(buf, tw, tid);
    {
        // fft_kept(u, H) with its constants hoisted: the shifted index of u lies in [lo, hi)
        const int hh = H / 2, lo = H / 4, hi = H - (H + 3) / 4;
        auto kept_y = [&](int u) {
            int i = u + hh;
            if (i >= H) i -= H;
            return i >= lo && i < hi;
        };
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int x = kx + c, nx = x == 0 ? 0 : W - x;  // (uniform: the x factors of the mask are per column)
            const bool fx = x < Wk && fft_kept(x, W), fnx = x < Wk && fft_kept(nx, W);
            for (int ky = tid; ky < H; ky += FFT_NT) {
                const int nky = ky == 0 ? 0 : H - ky;
                const int m = (int)(fx && kept_y(ky)) + (int)(fnx && kept_y(nky));
                // masked, normalised and conjugated: the inverse is conj(FFT(conj(.)))
                buf[c * H + ky] = cconj(cscale(buf[c * H + ky], 0.5f * (float)m * norm));
            }
        }
    }
    if constexpr (std::is_same<SP, NoPlan>::value) fft_lds(buf, H, NC, tw, H, rad, tid, FFT_NT);
    else fft_lds_s<SP, NC, FFT_NT>(buf, tw, tid);
    for (int y = tid; y < H; y += FFT_NT) {
        if (NC == 2) {
            const float2 a = cconj(buf[y]), b2 = cconj(buf[H + y]);
            col[(size_t)(TB / 2) * y] = make_float4(a.x, a.y, b2.x, b2.y);
        } else {
            colb[(size_t)TB * y] = cconj(buf[y]);
        }
    }
}

template <int RB, int NT, class SP = NoPlan>
__global__ void __launch_bounds__(NT, fft_rows_wpe(NT)) k_rows_inv(const float2* __restrict__ Tall, int H, int W, int Wk,
                                                                        FftFrames fr, HhsrRadices rad,
                                                                        const float2* __restrict__ twM, int twlen,
                                                                        const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int M = plan_len<SP>(W / 2), tid = threadIdx.x;
    W = 2 * M;
    float2* tw = fl;
    float2* buf = tw + ((twlen + 1) & ~1);  // 16-byte aligned
    batched_for<float2, NT>(twlen, tid, [&](int k) { return twM[k]; }, [&](int k, float2 v) { tw[k] = v; });
    const int nb = (H + RB - 1) / RB;
    for (int it = 0, vblk; (vblk = row_block<RB>(it, nb * fr.n)) >= 0; ++it) {  // persistent workgroups, see k_rows_fwd
    if (it) __syncthreads();
    const int frame = vblk / nb, blk = vblk - frame * nb;
    const float2* __restrict__ T = Tall + (size_t)frame * fr.tstride;
    float* __restrict__ dst = fr.dst[frame];
    const int y0 = blk * RB;
    const int nrows = min(RB, H - y0);
    // Z[k] = 1/2 [(X[k] + conj X[M-k]) + i conj(w_k) (X[k] - conj X[M-k])] with X = 0 above the kept band, stored
    // conjugated for the conj(FFT(conj(.))) inverse.  One thread builds the PAIR Z[k], Z[M-k] (k <= M/2) from X[k] and
    // X[M-k]: every kept bin is read from global memory exactly once, in the order it lies there — blocks of TB bins,
    // the workgroup's rows adjacent (the former element-wise loop read every bin twice, the second time in descending k).
    const int half = M / 2 + 1, nblk = (half + TB - 1) / TB;
    auto z_of = [&](float2 xk, float2 xm_, int k) {  // (xm_ = X[M-k], not yet conjugated)
        const float2 xm = cconj(xm_);
        const float2 s = cadd(xk, xm), d = mul_pi(cmul(cconj(twW[k]), csub(xk, xm)));
        return cconj(cscale(cadd(s, d), 0.5f));
    };
    if (TB == 2 && (M & 1) == 0 && nrows == RB) {
        // one thread per (bin pair, row): X[k], X[k+1] as one 16-byte load (the rows of the block are adjacent in T), no
        // integer division per bin; the mirrored bins X[M-k] are zero except around k = M/2
        for (int idx = tid; idx < nblk * RB; idx += NT) {
            const int b = idx / RB, rb = idx - b * RB, k0 = 2 * b, y = y0 + rb;  // (RB is a power of two)
            float4 x4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 < Wk) x4 = *reinterpret_cast<const float4*>(T + ((size_t)b * H + y) * 2);
            const float2 xa0 = make_float2(x4.x, x4.y);
            const float2 xa1 = k0 + 1 < Wk ? make_float2(x4.z, x4.w) : make_float2(0.f, 0.f);  // (bin Wk of an odd Wk is padding)
            float2* Zr = buf + __mul24(rb, M);
            float2 zk[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = k0 + e, mk = M - k;
                const float2 xa = e ? xa1 : xa0;
                float2 xb = make_float2(0.f, 0.f);
                if (mk == k) xb = xa;
                else if (mk < Wk) xb = T[t_index(mk, y, H)];
                zk[e] = z_of(xa, xb, k < M ? k : 0);
                if (k < half && mk != k && mk < M) Zr[mk] = z_of(xb, xa, mk);
            }
            if (k0 + 1 < half) *reinterpret_cast<float4*>(Zr + k0) = make_float4(zk[0].x, zk[0].y, zk[1].x, zk[1].y);
            else if (k0 < half) Zr[k0] = zk[0];
        }
    } else
    for (int idx = tid; idx < nblk * nrows * TB; idx += NT) {  // (batching these loads like batched_for: 2 us slower)
        const int b = idx / (nrows * TB), q = idx - b * (nrows * TB);
        const int rb = q / TB, k = b * TB + (q - rb * TB);
        if (k >= half) continue;
        const int mk = M - k;  // in M/2 .. M
        const float2 xa = k < Wk ? T[t_index(k, y0 + rb, H)] : make_float2(0.f, 0.f);
        const float2 xb = (mk < Wk && mk != k) ? T[t_index(mk, y0 + rb, H)] : (mk == k ? xa : make_float2(0.f, 0.f));
        {
            const float2 xk = xa, xm = cconj(xb);
            const float2 s = cadd(xk, xm), d = mul_pi(cmul(cconj(twW[k]), csub(xk, xm)));
            buf[rb * M + k] = cconj(cscale(cadd(s, d), 0.5f));
        }
        if (mk != k && mk < M) {
            const float2 xk = xb, xm = cconj(xa);
            const float2 s = cadd(xk, xm), d = mul_pi(cmul(cconj(twW[mk]), csub(xk, xm)));
            buf[rb * M + mk] = cconj(cscale(cadd(s, d), 0.5f));
        }
    }
    if constexpr (std::is_same<SP, NoPlan>::value) fft_lds(buf, M, nrows, tw, M, rad, tid, NT);
    else fft_lds_s<SP, RB, NT>(buf, tw, tid);
    if ((M & 1) == 0 && (W & 3) == 0) {
        const int Mh = M / 2;
        const float rMh = 1.0f / (float)Mh;
        float4* __restrict__ dst4 = reinterpret_cast<float4*>(dst + (size_t)y0 * W);  // rows are Mh float4 apart
        for (int idx = tid; idx < nrows * Mh; idx += NT) {  // x[2n] = Re z[n], x[2n+1] = -Im conj-stored z[n]
            int rb, n;
            split_row<RB>(idx, Mh, rMh, rb, n);
            const float4 z = reinterpret_cast<const float4*>(buf + __mul24(rb, M))[n];
            dst4[idx] = make_float4(z.x, -z.y, z.z, -z.w);
        }
    } else {
        for (int idx = tid; idx < nrows * M; idx += NT) {  // x[2n] = Re z[n], x[2n+1] = Im z[n]
            const int rb = idx / M, n = idx - rb * M;
            reinterpret_cast<float2*>(dst + (size_t)(y0 + rb) * W)[n] = cconj(buf[rb * M + n]);
        }
    }
    }
}

// ---- host side --------------------------------------------------------------------------------------------
static std::vector<float2> pass_twiddles(const HhsrRadices& rad);

// Radix schedule: fewest passes over the supported radices, then the smallest maximum radix (registers, idle
// lanes), then an odd / small first radix (the Ns = 1 pass stores with stride R: even R collide on LDS banks).
// HHSR_FFT_RADIX_MAX (experiments) caps the radix; 5 reproduces the original 5/4/3/2 schedule.
static const int k_radices[] = {16, 15, 14, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2};

static void radix_search(int n, int rmax, int cap, int nt, int depth, int* cur, int& best_n, int* best, int& best_max) {
    if (n == 1) {
        int mx = 0;
        for (int i = 0; i < depth; ++i) mx = cur[i] > mx ? cur[i] : mx;
        if (depth < best_n || (depth == best_n && mx < best_max)) {
            best_n = depth;
            best_max = mx;
            for (int i = 0; i < depth; ++i) best[i] = cur[i];
        }
        return;
    }
    if (depth + 1 > best_n || depth >= HHSR_MAX_RADICES) return;
    for (int r : k_radices) {
        if (r > rmax || n % r) continue;
        if (depth > 0 && r > cur[depth - 1]) continue;  // non-increasing: each multiset once
        if (cap / r > fft_maxit(r) * nt) continue;  // butterflies of this pass must fit the threads' registers
        cur[depth] = r;
        radix_search(n / r, rmax, cap, nt, depth + 1, cur, best_n, best, best_max);
    }
}

// nb: sequences transformed together by one workgroup of nt threads (every pass runs nb * n / R butterflies)
static bool factorize(int n, int nb, HhsrRadices& out, int nt = FFT_NT) {
    const char* e = getenv("HHSR_FFT_RADIX_MAX");
    const int rmax = e ? atoi(e) : 16;  // (radix 20 / 25 butterflies would exceed the 128-VGPR budget)
    int cur[HHSR_MAX_RADICES], best[HHSR_MAX_RADICES], best_n = HHSR_MAX_RADICES + 1, best_max = 1 << 30;
    radix_search(n, rmax, nb * n, nt, 0, cur, best_n, best, best_max);
    if (best_n > HHSR_MAX_RADICES) return false;
    // order: the radix with the fewest bank collisions at stride R first, then descending
    auto collide = [](int r) { int g = 2 * r, b = 64; while (b) { int t = g % b; g = b; b = t; } return g; };
    int first = 0;
    for (int i = 1; i < best_n; ++i)
        if (collide(best[i]) < collide(best[first]) || (collide(best[i]) == collide(best[first]) && best[i] > best[first]))
            first = i;
    static const int pow_min = getenv("HHSR_FFT_POW_MIN") ? atoi(getenv("HHSR_FFT_POW_MIN")) : HHSR_FFT_POW_MIN;
    out.pow_min = pow_min;
    out.n = 0;
    out.r[out.n++] = best[first];
    for (int i = 0; i < best_n; ++i)
        if (i != first) out.r[out.n++] = best[i];
    return out.n > 0;
}

static bool host_kept_fft(int u, int n) {
    int i = u + n / 2;
    if (i >= n) i -= n;
    return i >= n / 4 && i < n - (n + 3) / 4;
}

static float2* upload(const std::vector<float2>& h) {
    float2* d = nullptr;
    const int count = (int)h.size();
    if (hipMalloc((void**)&d, sizeof(float2) * count) != hipSuccess) return nullptr;
    if (count == 0) return nullptr;
    if (hipMemcpy(d, h.data(), sizeof(float2) * count, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    return d;
}

// exp(-2 pi i k / denom), k < count
static std::vector<float2> plain_twiddles(int count, double denom) {
    std::vector<float2> h(count);
    for (int k = 0; k < count; ++k) {
        const double a = -2.0 * M_PI * (double)k / denom;
        h[k] = make_float2((float)cos(a), (float)sin(a));
    }
    return h;
}

// per-pass tables, concatenated: pass p (radix R, sub-transform length Ns) holds (R-1)*Ns entries
// tw[(r-1)*Ns + k] = exp(-2 pi i r k / (Ns R))
static std::vector<float2> pass_twiddles(const HhsrRadices& rad) {
    std::vector<float2> h;
    int Ns = 1;
    for (int p = 0; p < rad.n; ++p) {
        const int R = rad.r[p];
        const bool pw = rad.pow_min > 0 && R <= HHSR_FFT_POW_RMAX && (R - 1) * Ns > rad.pow_min;  // only w^k: the kernel forms the powers
        for (int r = 1; r < (pw ? 2 : R); ++r)
            for (int k = 0; k < Ns; ++k) {
                const double a = -2.0 * M_PI * (double)r * (double)k / ((double)Ns * (double)R);
                h.push_back(make_float2((float)cos(a), (float)sin(a)));
            }
        Ns *= R;
    }
    return h;
}

// rows per workgroup of the row kernels: the first candidate whose LDS footprint leaves room for two workgroups
// per CU and whose passes fit the per-thread butterfly capacity
// First choice: one row per 256-thread workgroup when its passes fit 256 threads and row + twiddles fit 32 KB (five
// workgroups per CU; see FFT_NT_SMALL) — rows up to 4000 pixels.  HHSR_FFT_NT_ROWS=512 (experiments / tests) skips it.
static int pick_rb(int M, HhsrRadices& rad, int& nt) {
    const char* e = getenv("HHSR_FFT_RB");  // experiments
    const int forced = e ? atoi(e) : 0;
    const char* ent = getenv("HHSR_FFT_NT_ROWS");
    const int forced_nt = ent ? atoi(ent) : 0;
    if ((!forced || forced == 1) && forced_nt != FFT_NT && factorize(M, 1, rad, FFT_NT_SMALL) &&
        sizeof(float2) * (((pass_twiddles(rad).size() + 1) & ~(size_t)1) + (size_t)M) <= 32 * 1024) {
        nt = FFT_NT_SMALL;
        return 1;
    }
    nt = FFT_NT;
    const int cands[3] = {2, 4, 1};
    for (int c = 0; c < 3; ++c) {
        const int rb = cands[c];
        if (forced && rb != forced) continue;
        if (rb * (M / 2) >= 65536) continue;
        if (!factorize(M, rb, rad)) continue;
        if (sizeof(float2) * (pass_twiddles(rad).size() + (size_t)rb * M) > 76 * 1024 && rb > 1) continue;
        return rb;
    }
    return 0;
}

bool hhsr_fft_create(HhsrFft& f, int H, int W, int batch) {
    f = HhsrFft();
    if (batch < 1 || batch > HHSR_MAX_BATCH) return false;
    if (W % 2 || H < 2 || W < 4) return false;
    const int M = W / 2;
    if (M >= 65536 || H >= 65536) return false;
    f.rb = pick_rb(M, f.radM, f.nt_rows);
    if (!f.rb) return false;
    f.nc = 0;
    const char* enc = getenv("HHSR_FFT_NC");  // experiments / tests: force the columns-per-workgroup choice
    for (int nc = enc ? atoi(enc) : 2; nc >= 1 && !f.nc; --nc)  // two columns per workgroup unless their passes / LDS do not fit
        if (factorize(H, nc, f.radH) && sizeof(float2) * (pass_twiddles(f.radH).size() + (size_t)nc * H) <= 150 * 1024) f.nc = nc;
    if (!f.nc) return false;
    int Wk = 0;
    for (int x = 0; x <= M; ++x)
        if (host_kept_fft(x, W) || host_kept_fft(x == 0 ? 0 : W - x, W)) Wk = x + 1;
    if (Wk < 1 || Wk > M) return false;
    f.H = H;
    f.W = W;
    f.Wk = Wk;
    const std::vector<float2> hM = pass_twiddles(f.radM), hH = pass_twiddles(f.radH);
    f.twlenM = (int)hM.size();
    f.twlenH = (int)hH.size();
    f.lds_rows = sizeof(float2) * ((size_t)((f.twlenM + 1) & ~1) + (size_t)f.rb * M);
    f.lds_cols = sizeof(float2) * ((size_t)((f.twlenH + 1) & ~1) + (size_t)f.nc * H);
    if (f.lds_cols > 150 * 1024 || f.lds_rows > 150 * 1024) return false;
    const bool small = f.nt_rows == FFT_NT_SMALL;  // (only with rb = 1)
    const void* kf = small ? (const void*)k_rows_fwd<1, FFT_NT_SMALL> : f.rb == 4 ? (const void*)k_rows_fwd<4, FFT_NT>
                   : f.rb == 2 ? (const void*)k_rows_fwd<2, FFT_NT> : (const void*)k_rows_fwd<1, FFT_NT>;
    const void* ki = small ? (const void*)k_rows_inv<1, FFT_NT_SMALL> : f.rb == 4 ? (const void*)k_rows_inv<4, FFT_NT>
                   : f.rb == 2 ? (const void*)k_rows_inv<2, FFT_NT> : (const void*)k_rows_inv<1, FFT_NT>;
    if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rows) != hipSuccess ||
        hipFuncSetAttribute(ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rows) != hipSuccess ||
        hipFuncSetAttribute(f.nc == 2 ? (const void*)k_cols<2> : (const void*)k_cols<1>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_cols) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    f.twM = upload(hM);
    f.twH = upload(hH);
    f.twW = upload(plain_twiddles(M, (double)W));  // exp(-2 pi i k / W), k < M
    f.batch = batch;
    f.tstride = (size_t)((Wk + 7) / 8) * 8 * H;  // one spectrum per frame of a batch
    if (hipMalloc((void**)&f.T, sizeof(float2) * f.tstride * batch) != hipSuccess) f.T = nullptr;
    if (!f.twM || !f.twH || !f.twW || !f.T) {
        hhsr_fft_destroy(f);
        return false;
    }
    // static plans (HHSR_FFT_STATIC: bit 0 rows, bit 1 columns; 0: tests / A-B run the run-time passes)
    const char* es = getenv("HHSR_FFT_STATIC");
    const int allow = es ? atoi(es) : 3;
    f.static_rows = f.static_cols = 0;
#define HHSR_TRY_ROWS(ID, N, RB, NT, ...)                                                                                    \
    if ((allow & 1) && !f.static_rows && f.rb == RB && f.nt_rows == NT && SPlan<N, __VA_ARGS__>::matches(M, f.radM) &&       \
        hipFuncSetAttribute((const void*)k_rows_fwd<RB, NT, SPlan<N, __VA_ARGS__>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rows) == hipSuccess && \
        hipFuncSetAttribute((const void*)k_rows_inv<RB, NT, SPlan<N, __VA_ARGS__>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rows) == hipSuccess)   \
        f.static_rows = ID;
    HHSR_STATIC_ROWS(HHSR_TRY_ROWS)
#undef HHSR_TRY_ROWS
#define HHSR_TRY_COLS(ID, N, NC, NT, ...)                                                                                    \
    if ((allow & 2) && !f.static_cols && f.nc == NC && SPlan<N, __VA_ARGS__>::matches(H, f.radH) &&                          \
        hipFuncSetAttribute((const void*)k_cols<NC, SPlan<N, __VA_ARGS__>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_cols) == hipSuccess) \
        f.static_cols = ID;
    HHSR_STATIC_COLS(HHSR_TRY_COLS)
#undef HHSR_TRY_COLS
    (void)hipGetLastError();
    f.ok = true;
    return true;
}

void hhsr_fft_destroy(HhsrFft& f) {
    if (f.twM) (void)hipFree(f.twM);
    if (f.twH) (void)hipFree(f.twH);
    if (f.twW) (void)hipFree(f.twW);
    if (f.T) (void)hipFree(f.T);
    f = HhsrFft();
}

int hhsr_fft_lowpass(const HhsrFft& f, const float* const* srcs, float* const* dsts, int n, hipStream_t s) {
    // row kernels: at most one resident round of workgroups (3 per CU by their 48 kB of LDS, 5 per CU for the 256-thread
    // single-row workgroups with 32 kB), each walking several blocks
    static const int persist_env = getenv("HHSR_FFT_PERSIST") ? atoi(getenv("HHSR_FFT_PERSIST")) : -1;
    const bool small = f.nt_rows == FFT_NT_SMALL;
    const int persist = persist_env >= 0 ? persist_env : small ? 1280 : 768;
    // unnormalised inverse transforms multiply by (W/2) and H
    const float norm = (float)(1.0 / ((double)(f.W / 2) * (double)f.H));
    // Frames per round: the kept spectra of a round's frames live between the three kernels, and they should live in the
    // Infinity Cache (256 MB): 24 MB per 12 MP frame — four frames per round, the fewer launches win (57 us per frame against
    // 62 / 70 with two / one) —, 96 MB per 48 MP frame — ONE frame per round: 253 us per frame against 271 / 285 with two /
    // four, whose spectra go through HBM (tools/debug/fft_batch_probe.py).
    const size_t spectrum = sizeof(float2) * f.tstride, budget = (size_t)112 << 20;
    int per = (int)(budget / (spectrum ? spectrum : 1));
    per = per < 1 ? 1 : per > f.batch ? f.batch : per;
    static const int per_env = getenv("HHSR_FFT_ROUND") ? atoi(getenv("HHSR_FFT_ROUND")) : 0;  // (experiments)
    if (per_env > 0) per = per_env > f.batch ? f.batch : per_env;
    for (int n0 = 0; n0 < n; n0 += per) {  // (the plan holds f.batch spectra)
        FftFrames fr;
        fr.n = n - n0 < per ? n - n0 : per;
        fr.tstride = f.tstride;
        for (int k = 0; k < HHSR_MAX_BATCH; ++k) {
            fr.src[k] = srcs[n0 + (k < fr.n ? k : 0)];
            fr.dst[k] = dsts[n0 + (k < fr.n ? k : 0)];
        }
        const int nrb_all = hhsr_cdiv(f.H, f.rb) * fr.n;
        const int nrb = persist > 0 && persist < nrb_all ? persist : nrb_all;
#define ROWS_FWD(RB, NT) hipLaunchKernelGGL((k_rows_fwd<RB, NT>), dim3(nrb), dim3(NT), f.lds_rows, s, fr, f.H, f.W, f.T, f.Wk, \
                                            f.radM, f.twM, f.twlenM, f.twW)
#define ROWS_INV(RB, NT) hipLaunchKernelGGL((k_rows_inv<RB, NT>), dim3(nrb), dim3(NT), f.lds_rows, s, f.T, f.H, f.W, f.Wk, fr, \
                                            f.radM, f.twM, f.twlenM, f.twW)
#define HHSR_RUN_ROWS_FWD(ID, N, RB, NT, ...)                                                                                \
        if (f.static_rows == ID)                                                                                              \
            hipLaunchKernelGGL((k_rows_fwd<RB, NT, SPlan<N, __VA_ARGS__>>), dim3(nrb), dim3(NT), f.lds_rows, s, fr, f.H, f.W, f.T, \
                               f.Wk, f.radM, f.twM, f.twlenM, f.twW);                                                         \
        else
        HHSR_STATIC_ROWS(HHSR_RUN_ROWS_FWD)
#undef HHSR_RUN_ROWS_FWD
        if (small) ROWS_FWD(1, FFT_NT_SMALL); else if (f.rb == 4) ROWS_FWD(4, FFT_NT); else if (f.rb == 2) ROWS_FWD(2, FFT_NT);
        else ROWS_FWD(1, FFT_NT);
#define HHSR_RUN_COLS(ID, N, NC, NT, ...)                                                                                    \
        if (f.static_cols == ID)                                                                                              \
            hipLaunchKernelGGL((k_cols<NC, SPlan<N, __VA_ARGS__>>), dim3(((f.Wk + 63) / 64) * (64 / NC), fr.n), dim3(NT), f.lds_cols, \
                               s, f.T, f.tstride, f.H, f.W, f.Wk, f.radH, f.twH, f.twlenH, norm);                             \
        else
        HHSR_STATIC_COLS(HHSR_RUN_COLS)
#undef HHSR_RUN_COLS
        if (f.nc == 2)
            hipLaunchKernelGGL(k_cols<2>, dim3(((f.Wk + 63) / 64) * 32, fr.n), dim3(FFT_NT), f.lds_cols, s, f.T, f.tstride,
                               f.H, f.W, f.Wk, f.radH, f.twH, f.twlenH, norm);
        else
            hipLaunchKernelGGL(k_cols<1>, dim3(((f.Wk + 63) / 64) * 64, fr.n), dim3(FFT_NT), f.lds_cols, s, f.T, f.tstride,
                               f.H, f.W, f.Wk, f.radH, f.twH, f.twlenH, norm);
#define HHSR_RUN_ROWS_INV(ID, N, RB, NT, ...)                                                                                \
        if (f.static_rows == ID)                                                                                              \
            hipLaunchKernelGGL((k_rows_inv<RB, NT, SPlan<N, __VA_ARGS__>>), dim3(nrb), dim3(NT), f.lds_rows, s, f.T, f.H, f.W, f.Wk, \
                               fr, f.radM, f.twM, f.twlenM, f.twW);                                                           \
        else
        HHSR_STATIC_ROWS(HHSR_RUN_ROWS_INV)
#undef HHSR_RUN_ROWS_INV
        if (small) ROWS_INV(1, FFT_NT_SMALL); else if (f.rb == 4) ROWS_INV(4, FFT_NT); else if (f.rb == 2) ROWS_INV(2, FFT_NT);
        else ROWS_INV(1, FFT_NT);
#undef ROWS_FWD
#undef ROWS_INV
    }
    return hhsr_launch_status("hhsr_grey_lowpass");
}

// Fused in-LDS FFT low-pass for the grey image (Alg. 3, reference utils_image.py:82-100).
//
// The library route (hhsr_grey.hip) spends 9 full passes over the image per frame (row FFT, real-FFT
// post-processing + transpose, column FFT, transpose, mask, and the same backwards).  The mask keeps only
// |kx| <= W/4, |ky| <= H/4, so this file does the round trip in THREE kernels, each keeping a whole 1-D
// transform in LDS (Stockham autosort, radices 5/4/3/2, twiddles from an LDS table):
//   k_rows_fwd   2 rows per workgroup (simultaneously): real row -> half-length complex FFT -> real-FFT
//                post-processing; only the Wk = W/4 + 1 kept x-bins are written, blocked-transposed
//                (x-bins in blocks of 8, 64-byte runs);
//   k_cols       two kept columns per workgroup: forward FFT over y -> Hermitian mask m'(ky, kx) and the
//                1/(H W) normalisation -> inverse FFT, in place (24 MB in / out at 12 MP);
//   k_rows_inv   2 rows per workgroup: gather the kept bins, rebuild the half-length spectrum, inverse
//                FFT, write the real row.
// HBM / MALL traffic per 12 MP frame: 48 + 24 | 24 + 24 | 24 + 48 = 192 MB (library route: ~860 MB).
// Measured at 3000x4000: 171 us (library plans: 222 us); max abs error 4.8e-7 vs float64 (library 5.4e-7).
// The kernels are latency-bound on their ~16 barrier phases; fewer, fatter passes (radix 10/16 butterflies
// in registers) are the next step.
// Supported when W is even and W/2 and H factor into {2, 3, 5} and the LDS budgets fit; the caller falls
// back to the library plans otherwise.  Numerics: float32 butterflies, float64-computed twiddle tables.
#include "hhsr_common.h"
#include "hhsr_fft.h"
#include <math.h>
#include <vector>

// ---- complex helpers ------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)

// ---- forward R-point DFTs (w = exp(-2 pi i / R)) ------------------------------------------------------------
__device__ __forceinline__ void dft2(float2* v) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
__device__ __forceinline__ void dft3(float2* v) {
    const float s = 0.86602540378443864676f;
    const float2 t1 = cadd(v[1], v[2]);
    const float2 m1 = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
    const float2 d = cscale(mul_mi(csub(v[1], v[2])), s);  // -i s (b - c)
    v[0] = cadd(v[0], t1);
    v[1] = cadd(m1, d);
    v[2] = csub(m1, d);
}
__device__ __forceinline__ void dft4(float2* v) {
    const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    const float2 s13 = cadd(v[1], v[3]), d13 = mul_mi(csub(v[1], v[3]));  // -i (b - d)
    v[0] = cadd(s02, s13);
    v[2] = csub(s02, s13);
    v[1] = cadd(d02, d13);
    v[3] = csub(d02, d13);
}
__device__ __forceinline__ void dft5(float2* v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    const float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const float2 a = v[0];
    const float2 m1 = make_float2(a.x + c1 * t1.x + c2 * t2.x, a.y + c1 * t1.y + c2 * t2.y);
    const float2 m2 = make_float2(a.x + c2 * t1.x + c1 * t2.x, a.y + c2 * t1.y + c1 * t2.y);
    const float2 n1 = mul_mi(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));  // -i n1
    const float2 n2 = mul_mi(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));  // -i n2
    v[0] = cadd(a, cadd(t1, t2));
    v[1] = cadd(m1, n1);
    v[4] = csub(m1, n1);
    v[2] = cadd(m2, n2);
    v[3] = csub(m2, n2);
}

__constant__ int g_dbg_skip = 0;  // experiment knob: skip the last n passes (wrong results, timing only)

// One Stockham pass of radix R over NB independent length-N transforms stored `bstride` elements apart.
// twp: this pass's twiddles, twp[(r-1)*Ns + k] = exp(-2 pi i r k / (Ns R)) — contiguous in k, so the lanes of
// a wave (consecutive butterflies -> consecutive k) read consecutive LDS words (no bank conflicts).
template <int R>
__device__ __forceinline__ void stockham_pass(const float2* __restrict__ in, float2* __restrict__ out, int bstride,
                                              int NB, const float2* __restrict__ twp, int N, int Ns, int tid,
                                              int nt) {
    const int L = N / R;
    const float rNs = 1.0f / (float)Ns, rL = 1.0f / (float)L;
    for (int jj = tid; jj < NB * L; jj += nt) {
        const int bidx = (int)(((float)jj + 0.5f) * rL);  // exact floor for jj < 2^16 ... guarded by the host
        const int j = jj - bidx * L;
        const int q = (int)(((float)j + 0.5f) * rNs);
        const int k = j - q * Ns;
        const float2* ib = in + (size_t)bidx * bstride;
        float2* ob = out + (size_t)bidx * bstride;
        float2 v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = ib[j + r * L];
        if (Ns > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[r] = cmul(v[r], twp[(r - 1) * Ns + k]);
        }
        if (R == 2) dft2(v);
        else if (R == 3) dft3(v);
        else if (R == 4) dft4(v);
        else dft5(v);
        const int d = q * Ns * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) ob[d + r * Ns] = v[r];
    }
}

// Forward FFTs of NB length-N sequences: sequence b lives at buf + b*bstride in half `src_half` (0/1) of a
// double buffer whose halves are N elements apart.  Returns the half (0/1) holding the results.
// Every thread of the workgroup calls it; it starts and ends with a barrier.
__device__ __forceinline__ int fft_lds(float2* buf, int bstride, int NB, int src_half, const float2* tw, int N,
                                       const HhsrRadices& rad, int tid, int nt) {
    int Ns = 1, half = src_half, toff = 0;
    __syncthreads();
    for (int p = 0; p < rad.n - g_dbg_skip; ++p) {
        const int R = rad.r[p];
        const float2* in = buf + half * N;
        float2* out = buf + (half ^ 1) * N;
        if (R == 5) stockham_pass<5>(in, out, bstride, NB, tw + toff, N, Ns, tid, nt);
        else if (R == 4) stockham_pass<4>(in, out, bstride, NB, tw + toff, N, Ns, tid, nt);
        else if (R == 3) stockham_pass<3>(in, out, bstride, NB, tw + toff, N, Ns, tid, nt);
        else stockham_pass<2>(in, out, bstride, NB, tw + toff, N, Ns, tid, nt);
        __syncthreads();
        half ^= 1;
        toff += (R - 1) * Ns;
        Ns *= R;
    }
    return half;
}

__device__ __forceinline__ bool fft_kept(int u, int n) {
    int i = u + n / 2;
    if (i >= n) i -= n;
    return i >= n / 4 && i < n - (n + 3) / 4;
}

// Layout of the kept half spectrum T: x-bins in blocks of 8, element (kx, y) at ((kx/8)*H + y)*8 + kx%8 — the
// row kernels then move 64-byte runs (8 consecutive bins of one row) and a column is a 64-byte-strided walk
// whose cache lines are shared by the 8 columns of its block (placed on one XCD, see k_cols).
__device__ __forceinline__ size_t t_index(int kx, int y, int H) { return ((size_t)(kx >> 3) * H + y) * 8 + (kx & 7); }

constexpr int FFT_NT = 512;   // threads per workgroup, row kernels

// Row kernels: RB rows per workgroup, transformed SIMULTANEOUSLY (RB x fewer barriers, RB x more independent
// butterflies per thread).  LDS: tw[twlen] | RB x { half0[M] | half1[M] }  (float2 each).
template <int RB>
__global__ void __launch_bounds__(FFT_NT) k_rows_fwd(const float* __restrict__ src, int H, int W, float2* __restrict__ T,
                                                      int Wk, HhsrRadices rad, const float2* __restrict__ twM,
                                                      int twlen, const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int M = W / 2, tid = threadIdx.x;
    float2* tw = fl;
    float2* buf = tw + twlen;
    const int bstride = 2 * M;
    for (int k = tid; k < twlen; k += FFT_NT) tw[k] = twM[k];
    const int y0 = blockIdx.x * RB;
    const int nrows = min(RB, H - y0);
    for (int idx = tid; idx < nrows * M; idx += FFT_NT) {  // z[n] = x[2n] + i x[2n+1]
        const int rb = idx / M, n = idx - rb * M;
        buf[rb * bstride + n] = reinterpret_cast<const float2*>(src + (size_t)(y0 + rb) * W)[n];
    }
    const int h = fft_lds(buf, bstride, nrows, 0, tw, M, rad, tid, FFT_NT);
    // X[k] = 1/2 [(Z[k] + conj Z[M-k]) - i w_k (Z[k] - conj Z[M-k])],  w_k = exp(-2 pi i k / W); kept bins only,
    // written into the other half of the row's double buffer
    for (int idx = tid; idx < nrows * Wk; idx += FFT_NT) {
        const int rb = idx / Wk, k = idx - rb * Wk;
        const float2* Z = buf + rb * bstride + h * M;
        const float2 zk = Z[k == M ? 0 : k], zm = cconj(Z[k == 0 ? 0 : M - k]);
        const float2 s = cadd(zk, zm), d = mul_mi(cmul(twW[k], csub(zk, zm)));
        buf[rb * bstride + (h ^ 1) * M + k] = cscale(cadd(s, d), 0.5f);
    }
    __syncthreads();
    for (int idx = tid; idx < nrows * Wk; idx += FFT_NT) {  // blocked-transposed store, 64-byte runs
        const int rb = idx / Wk, k = idx - rb * Wk;
        T[t_index(k, y0 + rb, H)] = buf[rb * bstride + (h ^ 1) * M + k];
    }
}

// Column kernel: TWO adjacent kept columns per workgroup (one 16-byte load per row serves both), transformed
// simultaneously.  LDS: tw[twlen] | 2 x { half0[H] | half1[H] }
__global__ void __launch_bounds__(FFT_NT) k_cols(float2* __restrict__ T, int H, int W, int Wk, HhsrRadices rad,
                                                  const float2* __restrict__ twH, int twlen, float norm) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int tid = threadIdx.x;
    // workgroup b runs on XCD b % 8 (observed; locality only): give the 4 column pairs of one 64-byte block to
    // 4 consecutive workgroups of ONE XCD so that its L2 serves each cache line to all of them
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int kx = 8 * (xcd + 8 * (loc >> 2)) + 2 * (loc & 3);
    if (kx >= Wk) return;
    float2* tw = fl;
    float2* buf = tw + twlen;
    const int bstride = 2 * H;
    float4* col = reinterpret_cast<float4*>(T + ((size_t)(kx >> 3) * H) * 8 + (kx & 7));  // row y at col[4 y]
    for (int k = tid; k < twlen; k += FFT_NT) tw[k] = twH[k];
    for (int k = tid; k < H; k += FFT_NT) {
        const float4 v = col[(size_t)4 * k];
        buf[k] = make_float2(v.x, v.y);
        buf[bstride + k] = make_float2(v.z, v.w);
    }
    const int h = fft_lds(buf, bstride, 2, 0, tw, H, rad, tid, FFT_NT);
    for (int idx = tid; idx < 2 * H; idx += FFT_NT) {
        const int c = idx >= H, ky = idx - c * H;
        const int x = kx + c, nx = x == 0 ? 0 : W - x;
        const int nky = ky == 0 ? 0 : H - ky;
        const int m = x < Wk ? (int)(fft_kept(ky, H) && fft_kept(x, W)) + (int)(fft_kept(nky, H) && fft_kept(nx, W)) : 0;
        // masked, normalised and conjugated: the inverse is conj(FFT(conj(.)))
        buf[c * bstride + (h ^ 1) * H + ky] = cconj(cscale(buf[c * bstride + h * H + ky], 0.5f * (float)m * norm));
    }
    const int h2 = fft_lds(buf, bstride, 2, h ^ 1, tw, H, rad, tid, FFT_NT);
    for (int y = tid; y < H; y += FFT_NT) {
        const float2 a = cconj(buf[h2 * H + y]), b2 = cconj(buf[bstride + h2 * H + y]);
        col[(size_t)4 * y] = make_float4(a.x, a.y, b2.x, b2.y);
    }
}

template <int RB>
__global__ void __launch_bounds__(FFT_NT) k_rows_inv(const float2* __restrict__ T, int H, int W, int Wk,
                                                      float* __restrict__ dst, HhsrRadices rad,
                                                      const float2* __restrict__ twM, int twlen,
                                                      const float2* __restrict__ twW) {
    extern __shared__ __attribute__((aligned(16))) float2 fl[];
    const int M = W / 2, tid = threadIdx.x;
    float2* tw = fl;
    float2* buf = tw + twlen;
    const int bstride = 2 * M;
    for (int k = tid; k < twlen; k += FFT_NT) tw[k] = twM[k];
    const int y0 = blockIdx.x * RB;
    const int nrows = min(RB, H - y0);
    for (int idx = tid; idx < nrows * Wk; idx += FFT_NT) {  // kept bins -> half 1
        const int rb = idx / Wk, k = idx - rb * Wk;
        buf[rb * bstride + M + k] = T[t_index(k, y0 + rb, H)];
    }
    __syncthreads();
    // Z[k] = 1/2 [(X[k] + conj X[M-k]) + i conj(w_k) (X[k] - conj X[M-k])], X = 0 above the kept band; stored
    // conjugated (half 0) for the conj(FFT(conj(.))) inverse
    for (int idx = tid; idx < nrows * M; idx += FFT_NT) {
        const int rb = idx / M, k = idx - rb * M;
        const float2* X = buf + rb * bstride + M;
        const int mk = M - k;  // in 1..M
        const float2 xk = k < Wk ? X[k] : make_float2(0.f, 0.f);
        const float2 xm = mk < Wk ? cconj(X[mk]) : make_float2(0.f, 0.f);
        const float2 s = cadd(xk, xm), d = mul_pi(cmul(cconj(twW[k]), csub(xk, xm)));
        buf[rb * bstride + k] = cconj(cscale(cadd(s, d), 0.5f));
    }
    const int h = fft_lds(buf, bstride, nrows, 0, tw, M, rad, tid, FFT_NT);
    for (int idx = tid; idx < nrows * M; idx += FFT_NT) {  // x[2n] = Re z[n], x[2n+1] = Im z[n]
        const int rb = idx / M, n = idx - rb * M;
        reinterpret_cast<float2*>(dst + (size_t)(y0 + rb) * W)[n] = cconj(buf[rb * bstride + h * M + n]);
    }
}

// ---- host side --------------------------------------------------------------------------------------------
static bool factorize(int n, HhsrRadices& out) {
    out.n = 0;
    const int cand[4] = {5, 4, 3, 2};
    for (int c = 0; c < 4; ++c)
        while (n % cand[c] == 0 && n > 1) {
            if (out.n >= HHSR_MAX_RADICES) return false;
            out.r[out.n++] = cand[c];
            n /= cand[c];
        }
    return n == 1 && out.n > 0;
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
        for (int r = 1; r < R; ++r)
            for (int k = 0; k < Ns; ++k) {
                const double a = -2.0 * M_PI * (double)r * (double)k / ((double)Ns * (double)R);
                h.push_back(make_float2((float)cos(a), (float)sin(a)));
            }
        Ns *= R;
    }
    return h;
}

static int pick_rb(int M, int twlen) {
    const char* e = getenv("HHSR_FFT_RB");
    if (e) return atoi(e);
    const int cands[3] = {2, 1, 4};  // measured at 3000x4000: 2 rows per workgroup is the fastest
    for (int c = 0; c < 3; ++c)
        if (sizeof(float2) * ((size_t)twlen + (size_t)cands[c] * 2 * M) <= 150 * 1024 && cands[c] * (M / 2) < 65536)
            return cands[c];
    return 0;
}

bool hhsr_fft_create(HhsrFft& f, int H, int W) {
    f = HhsrFft();
    if (W % 2 || H < 2 || W < 4) return false;
    const int M = W / 2;
    if (M >= 65536 || H >= 65536) return false;
    if (!factorize(M, f.radM) || !factorize(H, f.radH)) return false;
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
    f.rb = pick_rb(M, f.twlenM);
    if (!f.rb) return false;
    f.lds_rows = sizeof(float2) * ((size_t)f.twlenM + (size_t)f.rb * 2 * M);
    f.lds_cols = sizeof(float2) * ((size_t)f.twlenH + (size_t)4 * H);
    if (f.lds_cols > 150 * 1024) return false;
    const void* kf = f.rb == 4 ? (const void*)k_rows_fwd<4> : f.rb == 2 ? (const void*)k_rows_fwd<2> : (const void*)k_rows_fwd<1>;
    const void* ki = f.rb == 4 ? (const void*)k_rows_inv<4> : f.rb == 2 ? (const void*)k_rows_inv<2> : (const void*)k_rows_inv<1>;
    if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rows) != hipSuccess ||
        hipFuncSetAttribute(ki, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_rows) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_cols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)f.lds_cols) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    f.twM = upload(hM);
    f.twH = upload(hH);
    f.twW = upload(plain_twiddles(M, (double)W));  // exp(-2 pi i k / W), k < M
    if (hipMalloc((void**)&f.T, sizeof(float2) * (size_t)((Wk + 7) / 8) * 8 * H) != hipSuccess) f.T = nullptr;
    if (!f.twM || !f.twH || !f.twW || !f.T) {
        hhsr_fft_destroy(f);
        return false;
    }
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

int hhsr_fft_lowpass(const HhsrFft& f, const float* src, float* dst, hipStream_t s) {
    static int dbg = -1;
    if (dbg < 0) {
        const char* e = getenv("HHSR_FFT_SKIP");
        dbg = e ? atoi(e) : 0;
        if (dbg) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_skip), &dbg, sizeof(int));
    }
    const int nrb = hhsr_cdiv(f.H, f.rb);
    // unnormalised inverse transforms multiply by (W/2) and H
    const float norm = (float)(1.0 / ((double)(f.W / 2) * (double)f.H));
#define ROWS_FWD(RB) hipLaunchKernelGGL(k_rows_fwd<RB>, dim3(nrb), dim3(FFT_NT), f.lds_rows, s, src, f.H, f.W, f.T, f.Wk, \
                                        f.radM, f.twM, f.twlenM, f.twW)
#define ROWS_INV(RB) hipLaunchKernelGGL(k_rows_inv<RB>, dim3(nrb), dim3(FFT_NT), f.lds_rows, s, f.T, f.H, f.W, f.Wk, dst, \
                                        f.radM, f.twM, f.twlenM, f.twW)
    if (f.rb == 4) ROWS_FWD(4); else if (f.rb == 2) ROWS_FWD(2); else ROWS_FWD(1);
    hipLaunchKernelGGL(k_cols, dim3(((f.Wk + 63) / 64) * 32), dim3(FFT_NT), f.lds_cols, s, f.T, f.H, f.W, f.Wk, f.radH, f.twH,
                       f.twlenH, norm);
    if (f.rb == 4) ROWS_INV(4); else if (f.rb == 2) ROWS_INV(2); else ROWS_INV(1);
#undef ROWS_FWD
#undef ROWS_INV
    return hhsr_launch_status("hhsr_grey_lowpass");
}

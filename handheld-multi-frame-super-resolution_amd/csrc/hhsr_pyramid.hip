// Grey-image low-pass masks and the Gaussian pyramid (reference utils_image.py:82-100, 360-391;
// alignment.py:27-37, 74-82).  HBM-bound streaming kernels; the decimating filter stages its input
// tile in LDS so every source pixel is fetched from L2/HBM ~1.4x instead of (4f+1)^2/f^2 times.
#include "hhsr_common.h"

// ---- low-pass masks --------------------------------------------------------------------------
// kept(u): un-shifted bin u survives the reference's zeroing of the fftshift-ed spectrum
// (rows [:n//4] and [-ceil(n/4):] removed; shifted index i = (u + n//2) mod n).
__device__ __forceinline__ bool lp_kept(int u, int n) {
    int i = u + n / 2;
    if (i >= n) i -= n;
    return i >= n / 4 && i < n - (n + 3) / 4;
}

// The spectrum may have any element strides (rocFFT's real transform returns a transposed layout
// through torch); `xfast` says which dimension is contiguous so that a wave reads consecutive bins.
__global__ void __launch_bounds__(256) k_lowpass_c2c(float2* __restrict__ spec, int H, int W, int64_t sy,
                                                      int64_t sx, int xfast) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const int x = xfast ? a : b, y = xfast ? b : a;
    if (x >= W || y >= H) return;
    if (!(lp_kept(y, H) && lp_kept(x, W))) spec[y * sy + x * sx] = make_float2(0.f, 0.f);
}

__global__ void __launch_bounds__(256) k_lowpass_r2c(float2* __restrict__ spec, int H, int W, int Wh, int64_t sy,
                                                      int64_t sx, int xfast) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    const int x = xfast ? a : b, y = xfast ? b : a;
    if (x >= Wh || y >= H) return;
    const int ny = y == 0 ? 0 : H - y, nx = x == 0 ? 0 : W - x;
    const int m = (int)(lp_kept(y, H) && lp_kept(x, W)) + (int)(lp_kept(ny, H) && lp_kept(nx, W));
    if (m == 2) return;
    float2 v = spec[y * sy + x * sx];
    const float s = 0.5f * (float)m;
    v.x *= s;
    v.y *= s;
    spec[y * sy + x * sx] = v;
}

extern "C" int hhsr_lowpass_mask_c2c(float* spec, int H, int W, int64_t stride_y, int64_t stride_x, void* stream) {
    HHSR_ARG(spec && H > 0 && W > 0 && stride_y > 0 && stride_x > 0);
    const int xfast = stride_x <= stride_y;
    const dim3 grid = xfast ? dim3(hhsr_cdiv(W, 256), H) : dim3(hhsr_cdiv(H, 256), W);
    hipLaunchKernelGGL(k_lowpass_c2c, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float2*>(spec), H, W,
                       stride_y, stride_x, xfast);
    HHSR_LAUNCHED();
}

extern "C" int hhsr_lowpass_mask_r2c(float* spec, int H, int W, int64_t stride_y, int64_t stride_x, void* stream) {
    HHSR_ARG(spec && H > 0 && W > 0 && stride_y > 0 && stride_x > 0);
    const int Wh = W / 2 + 1;
    const int xfast = stride_x <= stride_y;
    const dim3 grid = xfast ? dim3(hhsr_cdiv(Wh, 256), H) : dim3(hhsr_cdiv(H, 256), Wh);
    hipLaunchKernelGGL(k_lowpass_r2c, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float2*>(spec), H, W,
                       Wh, stride_y, stride_x, xfast);
    HHSR_LAUNCHED();
}

// ---- circular padding ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pad_circular(const float* __restrict__ src, int H, int W, int sp,
                                                       float* __restrict__ dst, int Hp, int Wp, int dp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wp) return;
    dst[(size_t)y * dp + x] = src[(size_t)(y % H) * sp + (x % W)];
}

extern "C" int hhsr_pad_circular(const float* src, int H, int W, int src_pitch, float* dst, int Hp, int Wp,
                                 int dst_pitch, void* stream) {
    HHSR_ARG(src && dst && H > 0 && W > 0 && Hp >= H && Wp >= W && src_pitch >= W && dst_pitch >= Wp);
    hipLaunchKernelGGL(k_pad_circular, dim3(hhsr_cdiv(Wp, 256), Hp), dim3(256), 0, (hipStream_t)stream, src, H, W,
                       src_pitch, dst, Hp, Wp, dst_pitch);
    HHSR_LAUNCHED();
}

// ---- separable Gaussian + decimation ------------------------------------------------------------
struct Taps {
    float g[HHSR_MAX_TAPS];
};

constexpr int GD_TX = 32, GD_TY = 16;  // output tile per 256-thread workgroup

// out[y][x] = sum_j g[j] * (sum_i g[i] * src[f*y+i][f*x+j]): rows first, then columns, both in float32 in
// tap order — the association of the reference's two chained valid conv2d calls.  F and the tap count
// NT = 4F+1 are compile-time so that the tile loads are issued back to back (a runtime-bound load->LDS
// loop serialises on memory latency) and the tap loops unroll.
struct GdFrames {  // blockIdx.z = frame of the batch
    const float* src[HHSR_MAX_BATCH];
    float* dst[HHSR_MAX_BATCH];
};

template <int F>
__global__ void __launch_bounds__(256) k_gauss_decimate(GdFrames fr, int H, int W, int sp,
                                                         int h2, int w2, int dp, Taps taps) {
    extern __shared__ float lds[];
    const float* __restrict__ src = fr.src[blockIdx.z];
    float* __restrict__ dst = fr.dst[blockIdx.z];
    constexpr int f = F, nt = 4 * F + 1;
    constexpr int IH = (GD_TY - 1) * f + nt, IW = (GD_TX - 1) * f + nt;
    constexpr int IWp = IW | 1;  // odd pitch: the strided column reads below stay <= 2-way conflicted
    float* s_in = lds;                  // [IH][IWp]
    float* s_tmp = lds + IH * IWp;      // [GD_TY][IWp]
    const int bid = xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);  // bands of tile rows per XCD
    const int ox0 = (bid % gridDim.x) * GD_TX, oy0 = (bid / gridDim.x) * GD_TY;
    const int ix0 = ox0 * f, iy0 = oy0 * f;
    const int tid = threadIdx.x;
    constexpr int NL = (IH * IW + 255) / 256;
    constexpr int CH = 8;  // loads in flight per thread
    for (int k0 = 0; k0 < NL; k0 += CH) {
        float v[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int p = tid + (k0 + k) * 256;
            const int r = p / IW, c = p - r * IW;
            const int y = min(iy0 + r, H - 1), x = min(ix0 + c, W - 1);
            v[k] = (k0 + k < NL && p < IH * IW) ? src[(size_t)y * sp + x] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int p = tid + (k0 + k) * 256;
            if (k0 + k < NL && p < IH * IW) s_in[(p / IW) * IWp + p % IW] = v[k];
        }
    }
    __syncthreads();
    for (int p = tid; p < GD_TY * IW; p += 256) {
        const int r = p / IW, c = p - r * IW;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < nt; ++i) acc += taps.g[i] * s_in[(r * f + i) * IWp + c];
        s_tmp[r * IWp + c] = acc;
    }
    __syncthreads();
    for (int p = tid; p < GD_TY * GD_TX; p += 256) {
        const int r = p / GD_TX, c = p - r * GD_TX;
        const int oy = oy0 + r, ox = ox0 + c;
        if (oy < h2 && ox < w2) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < nt; ++j) acc += taps.g[j] * s_tmp[r * IWp + c * f + j];
            dst[(size_t)oy * dp + ox] = acc;
        }
    }
}

extern "C" int hhsr_gauss_decimate_batch(const float* const* srcs, int n_frames, int H, int W, int src_pitch,
                                         float* const* dsts, int dst_pitch, int factor, const float* taps, int ntaps,
                                         void* stream) {
    HHSR_ARG(srcs && dsts && taps && n_frames >= 0 && H > 0 && W > 0 && src_pitch >= W);
    HHSR_ARG((factor == 2 || factor == 4) && ntaps == 4 * factor + 1);  // the reference's kernels: radius int(2f + 0.5)
    for (int n = 0; n < n_frames; ++n) HHSR_ARG(srcs[n] && dsts[n]);
    const int r = (ntaps - 1) / 2;
    const int h2 = (H - 2 * r) / factor, w2 = (W - 2 * r) / factor;
    HHSR_ARG(h2 > 0 && w2 > 0 && dst_pitch >= w2);
    Taps t;
    for (int i = 0; i < HHSR_MAX_TAPS; ++i) t.g[i] = i < ntaps ? taps[i] : 0.f;
    const int IH = (GD_TY - 1) * factor + ntaps, IWp = ((GD_TX - 1) * factor + ntaps) | 1;
    const size_t lds = (size_t)(IH + GD_TY) * IWp * sizeof(float);
    for (int n0 = 0; n0 < n_frames; n0 += HHSR_MAX_BATCH) {
        const int nb = n_frames - n0 < HHSR_MAX_BATCH ? n_frames - n0 : HHSR_MAX_BATCH;
        GdFrames fr;
        for (int k = 0; k < HHSR_MAX_BATCH; ++k) {
            fr.src[k] = srcs[n0 + (k < nb ? k : 0)];
            fr.dst[k] = dsts[n0 + (k < nb ? k : 0)];
        }
        const dim3 g(hhsr_cdiv(w2, GD_TX), hhsr_cdiv(h2, GD_TY), nb);
        if (factor == 2)
            hipLaunchKernelGGL(k_gauss_decimate<2>, g, dim3(256), lds, (hipStream_t)stream, fr, H, W, src_pitch, h2, w2,
                               dst_pitch, t);
        else
            hipLaunchKernelGGL(k_gauss_decimate<4>, g, dim3(256), lds, (hipStream_t)stream, fr, H, W, src_pitch, h2, w2,
                               dst_pitch, t);
    }
    HHSR_LAUNCHED();
}

extern "C" int hhsr_gauss_decimate(const float* src, int H, int W, int src_pitch, float* dst, int dst_pitch,
                                   int factor, const float* taps, int ntaps, void* stream) {
    HHSR_ARG(src && dst);
    return hhsr_gauss_decimate_batch(&src, 1, H, W, src_pitch, &dst, dst_pitch, factor, taps, ntaps, stream);
}

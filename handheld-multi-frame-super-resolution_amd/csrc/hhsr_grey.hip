// Alg. 3 grey image as a planned real FFT round trip (reference utils_image.py:82-100).
//
// The reference does fft2 -> fftshift -> zero 4 bands -> ifftshift -> ifft2 -> .real on a complex64 image
// (2 full complex transforms, 2 shift copies, 4 strided fills per frame).  Here: hipFFT (rocFFT) real-to-
// complex 2-D plan -> one in-place mask kernel on the half spectrum (the Hermitian mask m' = (m(k)+m(-k))/2
// with the 1/(H W) normalisation folded in) -> complex-to-real plan.  Half the transform work, no shift or
// normalisation passes, no temporaries beyond the plan's own spectrum buffer.
#include "hhsr_common.h"
#include "hhsr_fft.h"
#include <hipfft/hipfft.h>
#include <new>

struct GreyPlan {
    int H, W, Wh;
    int pruned, Wk;              // pruned: column transforms only on the Wk kept x-bins
    hipfftHandle r2c, c2r;       // 2-D plans (plain) or batched row plans (pruned)
    hipfftHandle col;            // pruned: batched strided column C2C plan
    float2* spec;                // [H][W/2+1]
    HhsrFft fft;                 // fused in-LDS transforms (HHSR_GREY_FUSED), when the sizes allow
};

__device__ __forceinline__ bool lp_kept2(int u, int n) {
    int i = u + n / 2;
    if (i >= n) i -= n;
    return i >= n / 4 && i < n - (n + 3) / 4;
}

__global__ void __launch_bounds__(256) k_lowpass_scale(float2* __restrict__ spec, int H, int W, int Wh, float norm) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wh) return;
    const int ny = y == 0 ? 0 : H - y, nx = x == 0 ? 0 : W - x;
    const int m = (int)(lp_kept2(y, H) && lp_kept2(x, W)) + (int)(lp_kept2(ny, H) && lp_kept2(nx, W));
    const size_t o = (size_t)y * Wh + x;
    if (m == 0) {
        spec[o] = make_float2(0.f, 0.f);
        return;
    }
    const float s = 0.5f * (float)m * norm;
    float2 v = spec[o];
    v.x *= s;
    v.y *= s;
    spec[o] = v;
}

// Transposed-pruned variant (HHSR_GREY_TPRUNED): the row transforms write / read the half spectrum
// TRANSPOSED ([kx][y], via hipFFT's advanced data layout), so the column transforms are contiguous batched
// C2C plans over the Wk kept x-bins only, the mask runs in the transposed domain and the two internal
// transposes of a 2-D plan between forward and inverse disappear.
__global__ void __launch_bounds__(256) k_lowpass_scale_t(float2* __restrict__ spec, int H, int W, int Wk, float norm) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x, x = blockIdx.y;  // spec[x][y]
    if (y >= H) return;
    const int ny = y == 0 ? 0 : H - y, nx = x == 0 ? 0 : W - x;
    const int m = (int)(lp_kept2(y, H) && lp_kept2(x, W)) + (int)(lp_kept2(ny, H) && lp_kept2(nx, W));
    const size_t o = (size_t)x * H + y;
    if (m == 0) {
        spec[o] = make_float2(0.f, 0.f);
        return;
    }
    const float s = 0.5f * (float)m * norm;
    float2 v = spec[o];
    v.x *= s;
    v.y *= s;
    spec[o] = v;
}

// Pruned variant: three quarters of the half spectrum are zeroed by the mask, so after the row transforms
// only the first Wk = (highest kept x-bin) + 1 columns need the column transforms (forward AND inverse):
//   rows R2C (batched) -> columns C2C forward on Wk columns -> mask -> columns C2C inverse -> rows C2R.
__global__ void __launch_bounds__(256) k_lowpass_scale_pruned(float2* __restrict__ spec, int H, int W, int Wh, int Wk,
                                                               float norm) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wh) return;
    const size_t o = (size_t)y * Wh + x;
    int m = 0;
    if (x < Wk) {
        const int ny = y == 0 ? 0 : H - y, nx = x == 0 ? 0 : W - x;
        m = (int)(lp_kept2(y, H) && lp_kept2(x, W)) + (int)(lp_kept2(ny, H) && lp_kept2(nx, W));
    }
    if (m == 0) {
        spec[o] = make_float2(0.f, 0.f);
        return;
    }
    const float s = 0.5f * (float)m * norm;
    float2 v = spec[o];
    v.x *= s;
    v.y *= s;
    spec[o] = v;
}

static bool host_kept(int u, int n) {
    int i = u + n / 2;
    if (i >= n) i -= n;
    return i >= n / 4 && i < n - (n + 3) / 4;
}

extern "C" int hhsr_grey_plan_create(int H, int W, int flags, void** plan_out) {
    HHSR_ARG(plan_out && H > 0 && W > 0);
    *plan_out = nullptr;
    GreyPlan* p = new (std::nothrow) GreyPlan();
    if (!p) {
        hhsr_set_error("hhsr_grey_plan_create: out of host memory");
        return -2;
    }
    p->H = H;
    p->W = W;
    p->Wh = W / 2 + 1;
    p->spec = nullptr;
    p->r2c = p->c2r = p->col = 0;
    p->pruned = flags & 1;
    p->Wk = 0;
    for (int x = 0; x < p->Wh; ++x)
        if (host_kept(x, W) || host_kept(x == 0 ? 0 : W - x, W)) p->Wk = x + 1;
    const int batch = (flags >> 8) & 0xff ? (flags >> 8) & 0xff : 1;  // HHSR_GREY_BATCH(n)
    if (batch > HHSR_MAX_BATCH) {
        hhsr_set_error("hhsr_grey_plan_create: batch %d > HHSR_MAX_BATCH", batch);
        delete p;
        return -1;
    }
    if ((flags & 4) && hhsr_fft_create(p->fft, H, W, batch)) {  // fused kernels: no library plans needed
        *plan_out = p;
        return 0;
    }
    hipError_t e = hipMalloc((void**)&p->spec, sizeof(float2) * (size_t)H * p->Wh);
    if (e != hipSuccess) {
        hhsr_set_error("hhsr_grey_plan_create: hipMalloc failed: %s", hipGetErrorString(e));
        delete p;
        return (int)e;
    }
    hipfftResult r1, r2;
    if ((flags & 2) && p->Wk > 0) {
        p->pruned = 2;
        int nW = W, nH = H;
        int iemb_r[1] = {W}, oemb_t[1] = {p->Wh}, emb_c[1] = {H};
        // rows: real [H][W] contiguous  <->  complex element (y, kx) at kx*H + y
        r1 = hipfftPlanMany(&p->r2c, 1, &nW, iemb_r, 1, W, oemb_t, H, 1, HIPFFT_R2C, H);
        r2 = r1 == HIPFFT_SUCCESS ? hipfftPlanMany(&p->c2r, 1, &nW, oemb_t, H, 1, iemb_r, 1, W, HIPFFT_C2R, H) : r1;
        if (r2 == HIPFFT_SUCCESS) r2 = hipfftPlanMany(&p->col, 1, &nH, emb_c, 1, H, emb_c, 1, H, HIPFFT_C2C, p->Wk);
    } else if (p->pruned && p->Wk > 0) {
        int nW = W, nH = H;
        int iemb_r[1] = {W}, oemb_r[1] = {p->Wh}, emb_c[1] = {H};
        r1 = hipfftPlanMany(&p->r2c, 1, &nW, iemb_r, 1, W, oemb_r, 1, p->Wh, HIPFFT_R2C, H);
        r2 = r1 == HIPFFT_SUCCESS ? hipfftPlanMany(&p->c2r, 1, &nW, oemb_r, 1, p->Wh, iemb_r, 1, W, HIPFFT_C2R, H) : r1;
        if (r2 == HIPFFT_SUCCESS)
            r2 = hipfftPlanMany(&p->col, 1, &nH, emb_c, p->Wh, 1, emb_c, p->Wh, 1, HIPFFT_C2C, p->Wk);
    } else {
        p->pruned = 0;
        r1 = hipfftPlan2d(&p->r2c, H, W, HIPFFT_R2C);
        r2 = r1 == HIPFFT_SUCCESS ? hipfftPlan2d(&p->c2r, H, W, HIPFFT_C2R) : r1;
    }
    if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) {
        hhsr_set_error("hhsr_grey_plan_create: hipfftPlan2d(%d, %d) failed (%d, %d)", H, W, (int)r1, (int)r2);
        if (r1 == HIPFFT_SUCCESS) hipfftDestroy(p->r2c);
        (void)hipFree(p->spec);
        delete p;
        return 1000 + (int)(r1 != HIPFFT_SUCCESS ? r1 : r2);
    }
    *plan_out = p;
    return 0;
}

extern "C" int hhsr_grey_plan_destroy(void* plan) {
    if (!plan) return 0;
    GreyPlan* p = static_cast<GreyPlan*>(plan);
    if (p->fft.ok) {
        hhsr_fft_destroy(p->fft);
        delete p;
        return 0;
    }
    hipfftDestroy(p->r2c);
    hipfftDestroy(p->c2r);
    if (p->col) hipfftDestroy(p->col);
    (void)hipFree(p->spec);
    delete p;
    return 0;
}

extern "C" int hhsr_grey_lowpass_batch(void* plan, const float* const* srcs, float* const* dsts, int n_frames, void* stream) {
    HHSR_ARG(plan && srcs && dsts && n_frames >= 0);
    for (int n = 0; n < n_frames; ++n) HHSR_ARG(srcs[n] && dsts[n]);
    GreyPlan* p = static_cast<GreyPlan*>(plan);
    if (p->fft.ok) return n_frames ? hhsr_fft_lowpass(p->fft, srcs, dsts, n_frames, (hipStream_t)stream) : 0;
    for (int n = 0; n < n_frames; ++n) {  // library plans: one spectrum buffer, frame by frame
        const int rc = hhsr_grey_lowpass(plan, srcs[n], dsts[n], stream);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int hhsr_grey_lowpass(void* plan, const float* src, float* dst, void* stream) {
    HHSR_ARG(plan && src && dst);
    GreyPlan* p = static_cast<GreyPlan*>(plan);
    hipStream_t s = (hipStream_t)stream;
    if (p->fft.ok) return hhsr_fft_lowpass(p->fft, &src, &dst, 1, s);
    hipfftComplex* sp = reinterpret_cast<hipfftComplex*>(p->spec);
    hipfftResult r = hipfftSetStream(p->r2c, s);
    if (r == HIPFFT_SUCCESS) r = hipfftSetStream(p->c2r, s);
    if (r == HIPFFT_SUCCESS && p->pruned) r = hipfftSetStream(p->col, s);
    if (r == HIPFFT_SUCCESS) r = hipfftExecR2C(p->r2c, const_cast<hipfftReal*>(src), sp);
    if (r == HIPFFT_SUCCESS && p->pruned) r = hipfftExecC2C(p->col, sp, sp, HIPFFT_FORWARD);
    if (r != HIPFFT_SUCCESS) {
        hhsr_set_error("hhsr_grey_lowpass: forward transform failed (%d)", (int)r);
        return 1000 + (int)r;
    }
    const float norm = 1.0f / ((float)p->H * (float)p->W);
    if (p->pruned == 2) {
        hipLaunchKernelGGL(k_lowpass_scale_t, dim3(hhsr_cdiv(p->H, 256), p->Wk), dim3(256), 0, s, p->spec, p->H, p->W,
                           p->Wk, norm);
        if (p->Wk < p->Wh)  // x-bins above the kept band feed the inverse row transforms as zeros
            (void)hipMemsetAsync(p->spec + (size_t)p->Wk * p->H, 0, sizeof(float2) * (size_t)(p->Wh - p->Wk) * p->H, s);
    } else if (p->pruned)
        hipLaunchKernelGGL(k_lowpass_scale_pruned, dim3(hhsr_cdiv(p->Wh, 256), p->H), dim3(256), 0, s, p->spec, p->H,
                           p->W, p->Wh, p->Wk, norm);
    else
        hipLaunchKernelGGL(k_lowpass_scale, dim3(hhsr_cdiv(p->Wh, 256), p->H), dim3(256), 0, s, p->spec, p->H, p->W,
                           p->Wh, norm);
    int rc = hhsr_launch_status("hhsr_grey_lowpass");
    if (rc) return rc;
    if (p->pruned) {
        r = hipfftExecC2C(p->col, sp, sp, HIPFFT_BACKWARD);
        if (r != HIPFFT_SUCCESS) {
            hhsr_set_error("hhsr_grey_lowpass: inverse column transform failed (%d)", (int)r);
            return 1000 + (int)r;
        }
    }
    r = hipfftExecC2R(p->c2r, reinterpret_cast<hipfftComplex*>(p->spec), dst);
    if (r != HIPFFT_SUCCESS) {
        hhsr_set_error("hhsr_grey_lowpass: inverse transform failed (%d)", (int)r);
        return 1000 + (int)r;
    }
    return 0;
}

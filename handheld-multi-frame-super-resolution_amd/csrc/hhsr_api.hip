// libhhsr_hip.so: version / error plumbing and the trivial element-wise entry points.
#include "hhsr_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void hhsr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* hhsr_version(void) { return "hhsr-hip 0.1 (gfx950)"; }
extern "C" const char* hhsr_last_error(void) { return g_err; }

// num /= den (reference utils.py:85-90); 0/0 stays NaN like the reference.
__global__ void __launch_bounds__(256) k_divide(float* __restrict__ num, const float* __restrict__ den, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n / 4;
    float4* n4p = reinterpret_cast<float4*>(num);
    const float4* d4p = reinterpret_cast<const float4*>(den);
    for (int64_t k = i; k < n4; k += stride) {
        float4 a = n4p[k];
        const float4 b = d4p[k];
        a.x /= b.x; a.y /= b.y; a.z /= b.z; a.w /= b.w;
        n4p[k] = a;
    }
    for (int64_t k = n4 * 4 + i; k < n; k += stride) num[k] = num[k] / den[k];
}

// A += B (reference utils.py:117-120; accumulated robustness, float32 here — SURVEY.md D15)
__global__ void __launch_bounds__(256) k_add(float* __restrict__ A, const float* __restrict__ B, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n / 4;
    float4* a4 = reinterpret_cast<float4*>(A);
    const float4* b4 = reinterpret_cast<const float4*>(B);
    for (int64_t k = i; k < n4; k += stride) {
        float4 a = a4[k];
        const float4 b = b4[k];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        a4[k] = a;
    }
    for (int64_t k = n4 * 4 + i; k < n; k += stride) A[k] += B[k];
}

// scalar forms: heads of / whole arrays whose pointers are not 16-byte aligned (views at odd offsets)
__global__ void __launch_bounds__(256) k_divide_scalar(float* __restrict__ num, const float* __restrict__ den, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) num[k] = num[k] / den[k];
}
__global__ void __launch_bounds__(256) k_add_scalar(float* __restrict__ A, const float* __restrict__ B, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) A[k] += B[k];
}

// Elements to handle with the scalar kernel before the float4 body may start: 0 for aligned pointers, < 4 when both
// pointers are off by the same number of floats, n (everything) when their misalignments differ.
static inline int64_t ew_head(const void* a, const void* b, int64_t n) {
    const uintptr_t ma = (uintptr_t)a & 15, mb = (uintptr_t)b & 15;
    if (ma == 0 && mb == 0) return 0;
    if (ma != mb || (ma & 3)) return n;
    const int64_t h = (int64_t)((16 - ma) / 4);
    return h < n ? h : n;
}

static inline int ew_grid(int64_t n) {
    int64_t b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (int)b;
}

extern "C" int hhsr_divide(float* num, const float* den, int64_t n, void* stream) {
    HHSR_ARG(num && den && n >= 0);
    HHSR_ARG(((uintptr_t)num & 3) == 0 && ((uintptr_t)den & 3) == 0);
    if (n == 0) return 0;
    const int64_t h = ew_head(num, den, n);
    if (h) hipLaunchKernelGGL(k_divide_scalar, dim3(ew_grid(4 * h)), dim3(256), 0, (hipStream_t)stream, num, den, h);
    if (n > h) hipLaunchKernelGGL(k_divide, dim3(ew_grid(n - h)), dim3(256), 0, (hipStream_t)stream, num + h, den + h, n - h);
    HHSR_LAUNCHED();
}

extern "C" int hhsr_add(float* A, const float* B, int64_t n, void* stream) {
    HHSR_ARG(A && B && n >= 0);
    HHSR_ARG(((uintptr_t)A & 3) == 0 && ((uintptr_t)B & 3) == 0);
    if (n == 0) return 0;
    const int64_t h = ew_head(A, B, n);
    if (h) hipLaunchKernelGGL(k_add_scalar, dim3(ew_grid(4 * h)), dim3(256), 0, (hipStream_t)stream, A, B, h);
    if (n > h) hipLaunchKernelGGL(k_add, dim3(ew_grid(n - h)), dim3(256), 0, (hipStream_t)stream, A + h, B + h, n - h);
    HHSR_LAUNCHED();
}

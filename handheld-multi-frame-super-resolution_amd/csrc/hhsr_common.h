// Shared host/device helpers for libhhsr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/hhsr.h"

#define HHSR_WAVE 64

// ---- error plumbing (thread local, no exceptions across the ABI) -------------------------------
void hhsr_set_error(const char* fmt, ...);

#define HHSR_ARG(cond)                                                                   \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            hhsr_set_error("%s: invalid argument: %s", __func__, #cond);                 \
            return -1;                                                                   \
        }                                                                                \
    } while (0)

static inline int hhsr_launch_status(const char* fn) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        hhsr_set_error("%s: launch failed: %s", fn, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
#define HHSR_LAUNCHED() return hhsr_launch_status(__func__)

static inline int hhsr_cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = HHSR_WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, HHSR_WAVE);
    return v;
}

// Block-wide sum of two values for blocks of NW waves; result valid in every thread.
// `sm` needs 2*NW floats.  Deterministic order.
template <int NW>
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sm) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & (HHSR_WAVE - 1), w = threadIdx.x / HHSR_WAVE;
    __syncthreads();  // protect sm from the previous use
    if (lane == 0) {
        sm[2 * w] = a;
        sm[2 * w + 1] = b;
    }
    __syncthreads();
    float ra = sm[0], rb = sm[1];
#pragma unroll
    for (int i = 1; i < NW; ++i) {
        ra += sm[2 * i];
        rb += sm[2 * i + 1];
    }
    a = ra;
    b = rb;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Python/Numba `max(0, z)`: returns z only when z > 0, so NaN -> 0 (reference quirk D10).
__device__ __forceinline__ double pymax0(double z) { return z > 0.0 ? z : 0.0; }
__device__ __forceinline__ float pymax0f(float z) { return z > 0.0f ? z : 0.0f; }

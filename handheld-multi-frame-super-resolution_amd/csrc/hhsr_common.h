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

// ---- wave64 reductions on the DPP path (no LDS crossbar, 6 VALU instructions) -----------------------------
// quad_perm xor-1 / xor-2, row_half_mirror, row_mirror leave every lane of a 16-lane row with the row total;
// row_bcast:15 / row_bcast:31 carry the totals across the four rows into lane 63, which is read back as a
// wave-uniform scalar.  Fixed association -> deterministic.
#define HHSR_DPP(v, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))

__device__ __forceinline__ float wave_sum_uniform(float v) {
    v += HHSR_DPP(v, 0xB1, 0xf);   // quad_perm [1,0,3,2]
    v += HHSR_DPP(v, 0x4E, 0xf);   // quad_perm [2,3,0,1]
    v += HHSR_DPP(v, 0x141, 0xf);  // row_half_mirror
    v += HHSR_DPP(v, 0x140, 0xf);  // row_mirror
    v += HHSR_DPP(v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    v += HHSR_DPP(v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float wave_min_uniform(float v) {
    // masked-out rows of the broadcast steps must see +inf, not 0: keep the old value with bound_ctrl = 0
#define HHSR_DPPM(v, ctrl, rmask) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (v)), __builtin_bit_cast(int, (v)), (ctrl), (rmask), 0xf, false))
    v = fminf(v, HHSR_DPPM(v, 0xB1, 0xf));
    v = fminf(v, HHSR_DPPM(v, 0x4E, 0xf));
    v = fminf(v, HHSR_DPPM(v, 0x141, 0xf));
    v = fminf(v, HHSR_DPPM(v, 0x140, 0xf));
    v = fminf(v, HHSR_DPPM(v, 0x142, 0xa));
    v = fminf(v, HHSR_DPPM(v, 0x143, 0xc));
#undef HHSR_DPPM
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// First minimum: smallest index among the lanes holding the minimal cost (wave-uniform result).
__device__ __forceinline__ int wave_argmin_first(float cost, int idx, float* min_cost = nullptr) {
    const float mn = wave_min_uniform(cost);
    // indices are < 2^24: carry them through the float min network exactly
    const float cand = (cost == mn) ? (float)idx : 3.0e38f;
    const int best = (int)wave_min_uniform(cand);
    if (min_cost) *min_cost = mn;
    return best;
}

// XCD-aware workgroup order.  The dispatcher places workgroup b on XCD b % 8 (observed; used for locality
// only, never for correctness): neighbouring workgroups, which share tile halos, land on different XCDs and
// each 4 MB L2 fetches the halo again.  This bijection gives every XCD one contiguous range of logical ids.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, rem = nblk & 7;
    return xcd * q + min(xcd, rem) + loc;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Python/Numba `max(0, z)`: returns z only when z > 0, so NaN -> 0 (reference quirk D10).
__device__ __forceinline__ double pymax0(double z) { return z > 0.0 ? z : 0.0; }
__device__ __forceinline__ float pymax0f(float z) { return z > 0.0f ? z : 0.0f; }

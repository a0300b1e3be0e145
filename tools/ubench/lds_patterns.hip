// LDS access-pattern micro-benchmark for k_merge_x2's layouts (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_patterns.hip -o gpurun_out/lds_patterns
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITERS 4096
__device__ __forceinline__ void sink(float v, float* out) { if (v == 1234.5f) *out = v; }

// lane = li * 8 + lj ; address (dwords) = li * ROWSTEP + COLSTEP * lj
template <int ROWSTEP, int COLSTEP>
__global__ void k_b32(float* out) {
    __shared__ float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane >> 3, lj = lane & 7;
    const volatile float* p = s + li * ROWSTEP + COLSTEP * lj;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) acc += p[(i & 7) * 2];
    sink(acc, out);
}
template <int ROWSTEP, int COLSTEP>
__global__ void k_b64(float* out) {
    __shared__ __align__(16) float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane >> 3, lj = lane & 7;
    const float* p = s + li * ROWSTEP + COLSTEP * lj;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        float2 v;
        asm volatile("ds_read_b64 %0, %1 offset:0\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(p + (i & 7) * 2)) : "memory");
        acc += v.x + v.y;
    }
    sink(acc, out);
}
template <int ROWSTEP, int COLSTEP>
__global__ void k_b128(float* out) {
    __shared__ __align__(16) float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane >> 3, lj = lane & 7;
    const float* p = s + li * ROWSTEP + COLSTEP * lj;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        float4 v;
        asm volatile("ds_read_b128 %0, %1 offset:0\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(p + (i & 3) * 4)) : "memory");
        acc += v.x + v.w;
    }
    sink(acc, out);
}
template <int ROWSTEP, int COLSTEP>
__global__ void k_r2b64(float* out) {
    __shared__ __align__(16) float s[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane >> 3, lj = lane & 7;
    const float* p = s + li * ROWSTEP + COLSTEP * lj;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
        float4 v;
        asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(p + (i & 3) * 4)) : "memory");
        acc += v.x + v.w;
    }
    sink(acc, out);
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    const dim3 g(1024), b(256);
    // raw window: pitch 24, rows stride 2 -> 48; columns stride 2
    hipLaunchKernelGGL((k_b32<48, 2>), g, b, 0, 0, out);      // third-column dword reads (expected 2-way)
    hipLaunchKernelGGL((k_b32<8, 1>), g, b, 0, 0, out);       // consecutive dwords (conflict-free reference)
    hipLaunchKernelGGL((k_b64<48, 2>), g, b, 0, 0, out);      // aligned pairs, rows stride 2 (k_merge_x2)
    hipLaunchKernelGGL((k_b64<16, 2>), g, b, 0, 0, out);      // fully contiguous b64
    hipLaunchKernelGGL((k_b128<96, 4>), g, b, 0, 0, out);     // covariance cells, pitch 24 float4
    hipLaunchKernelGGL((k_b128<32, 4>), g, b, 0, 0, out);     // fully contiguous b128
    hipLaunchKernelGGL((k_b128<48, 4>), g, b, 0, 0, out);     // pitch 12 float4
    hipLaunchKernelGGL((k_r2b64<48, 2>), g, b, 0, 0, out);    // compiler-merged pairs
    hipLaunchKernelGGL((k_r2b64<32, 4>), g, b, 0, 0, out);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}

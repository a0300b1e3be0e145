// VALU issue rate against OCCUPANCY (gfx950): wave64 instructions per cycle and SIMD with k = 1 .. 8 waves per SIMD
// (one 256-thread workgroup = one wave per SIMD; dynamic LDS sized so that exactly k workgroups fit a CU) for
// instruction streams with 8, 2 and 1 independent chains per lane — does a kernel that lives at 2-3 waves per SIMD
// (k_merge_xs<3>: 72 accumulators) lose VALU throughput to occupancy by itself, before any LDS / barrier latency?
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_occupancy.hip -o tools/ubench/valu_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096

extern __shared__ float dyn[];

#define KERNEL(NAME, ASM)                                                                  \
    __global__ void __launch_bounds__(256) NAME(float* out) {                             \
        float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;       \
        float v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;                   \
        float c = 0.999f, d = 1e-3f;                                                       \
        if (out == nullptr) dyn[threadIdx.x] = v0;                                         \
        for (int i = 0; i < ITERS; ++i) {                                                  \
            asm volatile(ASM ASM ASM ASM                                                   \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                         : "v"(c), "v"(d));                                                \
        }                                                                                  \
        float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                   \
        if (s == 1234.5f) *out = s;                                                        \
    }

// 8 instructions per ASM block each
#define A_FMA8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define A_FMA2 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
#define A_FMA1 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
// the merge kernels' tap: fma, fma, exp (clamped), fmac, add — two taps interleaved (12 instructions; counted as 12)
#define A_TAP2 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32_e64 %1, %0 clamp\n v_exp_f32_e64 %5, %4 clamp\n v_fmac_f32 %2, %1, %8\n v_fmac_f32 %6, %5, %8\n v_add_f32 %3, %3, %1\n v_add_f32 %7, %7, %5\n"
// ... one tap at a time (fully dependent chain of 5)
#define A_TAP1 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_exp_f32_e64 %1, %0 clamp\n v_fmac_f32 %2, %1, %8\n v_add_f32 %3, %3, %1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32_e64 %5, %4 clamp\n v_fmac_f32 %6, %5, %8\n v_add_f32 %7, %7, %5\n"
KERNEL(k_fma8, A_FMA8)
KERNEL(k_fma2, A_FMA2)
KERNEL(k_fma1, A_FMA1)
KERNEL(k_tap2, A_TAP2)
KERNEL(k_tap1, A_TAP1)

typedef void (*kern_t)(float*);

static double run(kern_t k, int waves_per_simd, int ninstr_per_block, float* out, int cus) {
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = cus * waves_per_simd;  // exactly one resident round
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, 0, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves x ITERS x 4 blocks x ninstr instructions
    const double insts = (double)waves_per_simd * ITERS * 4.0 * ninstr_per_block;
    const double cycles = ms * 1e-3 * 2.4e9;
    return cycles / insts;  // cycles per wave64 instruction per SIMD
}

int main() {
    float* out;
    (void)hipMalloc((void**)&out, 4);
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("# cycles (at 2.4 GHz) per wave64 instruction per SIMD; %d CUs; one resident round of 256-thread workgroups\n", cus);
    printf("%-28s", "waves per SIMD:");
    const int occ[] = {1, 2, 3, 4, 6, 8};
    for (int o : occ) printf("%8d", o);
    printf("\n");
    struct { const char* name; kern_t k; int n; } ks[] = {
        {"fma, 8 chains", k_fma8, 8}, {"fma, 2 chains", k_fma2, 8}, {"fma, 1 chain", k_fma1, 8},
        {"tap x2 interleaved", k_tap2, 10}, {"tap, dependent", k_tap1, 10}};
    for (auto& k : ks) {
        printf("%-28s", k.name);
        for (int o : occ) printf("%8.2f", run(k.k, o, k.n, out, cus));
        printf("\n");
    }
    return 0;
}

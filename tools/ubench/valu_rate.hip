// VALU issue-rate micro-benchmark (gfx950): cycles per wave64 instruction per SIMD for the instruction mix of the merge
// kernel.  8 independent chains per lane, 8 waves per SIMD, so dependencies and latency are hidden; time by hipEvent.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, ASM)                                                                  \
    __global__ void __launch_bounds__(256) NAME(float* out) {                             \
        float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;       \
        float v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;                   \
        float c = 0.999f, d = 1e-3f;                                                       \
        for (int i = 0; i < ITERS; ++i) {                                                  \
            asm volatile(ASM ASM ASM ASM                                                   \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                         : "v"(c), "v"(d));                                                \
        }                                                                                  \
        float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                   \
        if (s == 1234.5f) *out = s;                                                        \
    }

#define A_FMA  "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define A_FMAC "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
#define A_MUL  "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
#define A_ADD  "v_add_f32 %0, %0, %9\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %9\n v_add_f32 %3, %3, %9\n v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_add_f32 %7, %7, %9\n"
#define A_MIN  "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
#define A_MIN3 "v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
#define A_EXP  "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
#define A_RCP  "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
#define A_MOV  "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
#define A_CND  "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
#define A_IADD "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
// mixed: the tap body of the merge kernel: 2 fma + min + exp + fma + add
#define A_TAP  "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_min_f32 %0, %0, %8\n v_exp_f32 %1, %0\n v_fmac_f32 %2, %1, %8\n v_add_f32 %3, %3, %1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_min_f32 %4, %4, %8\n v_exp_f32 %5, %4\n v_fmac_f32 %6, %5, %8\n v_add_f32 %7, %7, %5\n"

#define A_CND64 "v_cndmask_b32 %0, %0, %8, s[6:7]\n v_cndmask_b32 %1, %1, %8, s[6:7]\n v_cndmask_b32 %2, %2, %8, s[6:7]\n v_cndmask_b32 %3, %3, %8, s[6:7]\n v_cndmask_b32 %4, %4, %8, s[6:7]\n v_cndmask_b32 %5, %5, %8, s[6:7]\n v_cndmask_b32 %6, %6, %8, s[6:7]\n v_cndmask_b32 %7, %7, %8, s[6:7]\n"
#define A_CMP  "v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n"
#define A_MAX  "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n"
#define A_FLOOR "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n"
#define A_CVT  "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n"
#define A_CLAMPEXP "v_exp_f32_e64 %0, %0 clamp\n v_exp_f32_e64 %1, %1 clamp\n v_exp_f32_e64 %2, %2 clamp\n v_exp_f32_e64 %3, %3 clamp\n v_exp_f32_e64 %4, %4 clamp\n v_exp_f32_e64 %5, %5 clamp\n v_exp_f32_e64 %6, %6 clamp\n v_exp_f32_e64 %7, %7 clamp\n"
#define A_FMACLAMP "v_fma_f32 %0, %0, %8, %9 clamp\n v_fma_f32 %1, %1, %8, %9 clamp\n v_fma_f32 %2, %2, %8, %9 clamp\n v_fma_f32 %3, %3, %8, %9 clamp\n v_fma_f32 %4, %4, %8, %9 clamp\n v_fma_f32 %5, %5, %8, %9 clamp\n v_fma_f32 %6, %6, %8, %9 clamp\n v_fma_f32 %7, %7, %8, %9 clamp\n"
#define A_MULLO "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
// the CLAMP form of the tap (2 fma + clamped exp + fmac + add), each instruction depending on the one before it ...
#define A_TAP5 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_exp_f32_e64 %1, %0 clamp\n v_fmac_f32 %2, %1, %8\n v_add_f32 %3, %3, %1\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_exp_f32_e64 %5, %4 clamp\n v_fmac_f32 %6, %5, %8\n v_add_f32 %7, %7, %5\n"
// ... and four taps batched by instruction class (8 fma, 4 exp, 4 fmac, 4 add): does the ORDER matter at 8 waves per SIMD?
#define A_TAP5B "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_exp_f32_e64 %0, %0 clamp\n v_exp_f32_e64 %1, %1 clamp\n v_exp_f32_e64 %2, %2 clamp\n v_exp_f32_e64 %3, %3 clamp\n v_fmac_f32 %4, %0, %8\n v_fmac_f32 %5, %1, %8\n v_fmac_f32 %4, %2, %8\n v_fmac_f32 %5, %3, %8\n v_add_f32 %6, %6, %0\n v_add_f32 %7, %7, %1\n v_add_f32 %6, %6, %2\n v_add_f32 %7, %7, %3\n"
KERNEL(k_tap5, A_TAP5)
KERNEL(k_tap5b, A_TAP5B)
KERNEL(k_cnd64, A_CND64)
KERNEL(k_cmp, A_CMP)
KERNEL(k_max, A_MAX)
KERNEL(k_floor, A_FLOOR)
KERNEL(k_cvt, A_CVT)
KERNEL(k_clampexp, A_CLAMPEXP)
KERNEL(k_fmaclamp, A_FMACLAMP)
KERNEL(k_mullo, A_MULLO)
KERNEL(k_fma, A_FMA)
KERNEL(k_fmac, A_FMAC)
KERNEL(k_mul, A_MUL)
KERNEL(k_add, A_ADD)
KERNEL(k_min, A_MIN)
KERNEL(k_min3, A_MIN3)
KERNEL(k_exp, A_EXP)
KERNEL(k_rcp, A_RCP)
KERNEL(k_mov, A_MOV)
KERNEL(k_cnd, A_CND)
KERNEL(k_iadd, A_IADD)
KERNEL(k_tap, A_TAP)


#define KERNEL64(NAME, ASM)                                                                \
    __global__ void __launch_bounds__(256) NAME(float* out) {                             \
        double v0 = threadIdx.x * 1e-3 + 1.0, v1 = v0 + 1., v2 = v0 + 2., v3 = v0 + 3.;     \
        double v4 = v0 + 4., v5 = v0 + 5., v6 = v0 + 6., v7 = v0 + 7.;                      \
        double c = 0.999, d = 1e-3;                                                        \
        for (int i = 0; i < ITERS; ++i) {                                                  \
            asm volatile(ASM ASM ASM ASM                                                   \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) \
                         : "v"(c), "v"(d));                                                \
        }                                                                                  \
        double s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                  \
        if (s == 1234.5) *out = (float)s;                                                  \
    }
#define R8(OP, ARGS) OP " %0, %0" ARGS "\n " OP " %1, %1" ARGS "\n " OP " %2, %2" ARGS "\n " OP " %3, %3" ARGS "\n " OP " %4, %4" ARGS "\n " OP " %5, %5" ARGS "\n " OP " %6, %6" ARGS "\n " OP " %7, %7" ARGS "\n "
KERNEL64(k_mul64, R8("v_mul_f64", ", %8"))
KERNEL64(k_add64, R8("v_add_f64", ", %9"))
KERNEL64(k_fma64, R8("v_fma_f64", ", %8, %9"))
KERNEL64(k_rcp64, R8("v_rcp_f64", ""))
KERNEL64(k_rsq64, R8("v_rsq_f64", ""))
KERNEL64(k_sqrt64, R8("v_sqrt_f64", ""))
KERNEL64(k_max64, R8("v_max_f64", ", %8"))
KERNEL(k_sqrt32, R8("v_sqrt_f32", ""))
// conversions: f32 <-> f64 round trip (2 instructions per chain element)
__global__ void __launch_bounds__(256) k_cvt64(float* out) {
    float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    double d0, d1, d2, d3;
    for (int i = 0; i < ITERS; ++i) {
#define CV "v_cvt_f64_f32 %4, %0\n v_cvt_f64_f32 %5, %1\n v_cvt_f64_f32 %6, %2\n v_cvt_f64_f32 %7, %3\n v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7\n"
        asm volatile(CV CV CV CV : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3));
    }
    if (v0 + v1 + v2 + v3 == 1234.5f) *out = v0;
}


// compare + select pairs as compilers emit them: through VCC (VOP2 v_cndmask) and through an SGPR pair (VOP3)
#define A_CMPSEL_VCC "v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
#define A_CMPSEL_SGPR "v_cmp_gt_f32 s[6:7], %0, %8\n v_cndmask_b32 %0, %0, %9, s[6:7]\n v_cmp_gt_f32 s[8:9], %1, %8\n v_cndmask_b32 %1, %1, %9, s[8:9]\n v_cmp_gt_f32 s[6:7], %2, %8\n v_cndmask_b32 %2, %2, %9, s[6:7]\n v_cmp_gt_f32 s[8:9], %3, %8\n v_cndmask_b32 %3, %3, %9, s[8:9]\n"
// independent compares first, selects later (scheduled apart)
#define A_CMPSEL_APART "v_cmp_gt_f32 s[6:7], %0, %8\n v_cmp_gt_f32 s[8:9], %1, %8\n v_cmp_gt_f32 s[10:11], %2, %8\n v_cmp_gt_f32 s[12:13], %3, %8\n v_cndmask_b32 %0, %0, %9, s[6:7]\n v_cndmask_b32 %1, %1, %9, s[8:9]\n v_cndmask_b32 %2, %2, %9, s[10:11]\n v_cndmask_b32 %3, %3, %9, s[12:13]\n"
KERNEL(k_cmpsel_vcc, A_CMPSEL_VCC)
KERNEL(k_cmpsel_sgpr, A_CMPSEL_SGPR)
KERNEL(k_cmpsel_apart, A_CMPSEL_APART)

// packed: two floats per register pair
__global__ void __launch_bounds__(256) k_pkfma(float* out) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a0 = {threadIdx.x * 1e-3f, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    v2 c = {0.999f, 0.998f}, d = {1e-3f, 2e-3f};
    for (int i = 0; i < ITERS; ++i) {
#define PK "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
        asm volatile(PK PK PK PK : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    }
    v2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x + s.y == 1234.5f) *out = s.x;
}

template <class K>
static void run(const char* name, K k, int per_iter, float* out) {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITERS * per_iter;          // wave-instructions
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / winstr;                 // cycles per wave-instruction per SIMD @ 2.4 GHz
    printf("%-8s %8.3f ms  %6.2f cycles / wave64-instruction / SIMD (at 2.4 GHz)\n", name, ms, cyc);
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    run("fma", k_fma, 32, out);
    run("fmac", k_fmac, 32, out);
    run("mul", k_mul, 32, out);
    run("add", k_add, 32, out);
    run("min", k_min, 32, out);
    run("min3", k_min3, 32, out);
    run("exp", k_exp, 32, out);
    run("rcp", k_rcp, 32, out);
    run("mov", k_mov, 32, out);
    run("cndmask", k_cnd, 32, out);
    run("iadd", k_iadd, 32, out);
    run("pk_fma", k_pkfma, 32, out);
    run("tap mix", k_tap, 48, out);
    run("tap5 dependent order", k_tap5, 40, out);
    run("tap5 batched by class", k_tap5b, 80, out);
    run("cnd e64", k_cnd64, 32, out);
    run("cmp", k_cmp, 32, out);
    run("max", k_max, 32, out);
    run("floor", k_floor, 32, out);
    run("cvt_i32", k_cvt, 32, out);
    run("exp clamp", k_clampexp, 32, out);
    run("fma clamp", k_fmaclamp, 32, out);
    run("mul_lo", k_mullo, 32, out);
    run("cmp+sel vcc", k_cmpsel_vcc, 32, out);
    run("cmp+sel sgpr", k_cmpsel_sgpr, 32, out);
    run("cmp,sel apart", k_cmpsel_apart, 32, out);
    run("sqrt f32", k_sqrt32, 32, out);
    run("mul f64", k_mul64, 32, out);
    run("add f64", k_add64, 32, out);
    run("fma f64", k_fma64, 32, out);
    run("max f64", k_max64, 32, out);
    run("rcp f64", k_rcp64, 32, out);
    run("rsq f64", k_rsq64, 32, out);
    run("sqrt f64", k_sqrt64, 32, out);
    run("cvt f64", k_cvt64, 32, out);
    return 0;
}

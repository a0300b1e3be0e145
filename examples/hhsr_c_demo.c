/* Plain-C client of libhhsr_hip.so: no Python, no torch — the drop-in boundary is the C ABI of include/hhsr.h.
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/hhsr_c_demo.c \
 *       -Lhandheld-multi-frame-super-resolution_amd/handheld_super_resolution -lhhsr_hip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN' -o hhsr_c_demo
 *
 * Normalises a synthetic uint16 Bayer frame (hhsr_normalize_raw_u16), builds one Gaussian pyramid level
 * (hhsr_gauss_decimate) and checks both against the same arithmetic done on the host.  Exit code 0 = match. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "hhsr.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_HHSR(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, hhsr_last_error()); return 3; } } while (0)

int main(void) {
    enum { H = 64, W = 96, F = 2, NT = 4 * F + 1 };
    static uint16_t counts[H * W];
    static float norm_host[H * W], norm_dev[H * W];
    const uint8_t cfa[4] = {0, 1, 1, 2};
    const double black[3] = {64, 64, 64}, wb[3] = {1.9, 1.0, 1.6};
    const double white = 4095;
    for (int i = 0; i < H * W; ++i) counts[i] = (uint16_t)(64 + (i * 2654435761u >> 20) % 4000);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int c = cfa[(y & 1) * 2 + (x & 1)];
            const float v = ((float)counts[y * W + x] - (float)black[c]) / (float)(white - black[c]);
            norm_host[y * W + x] = v * (float)(wb[c] / wb[1]);
        }
    printf("%s\n", hhsr_version());
    uint16_t* d_counts;
    float *d_norm, *d_lvl;
    CHECK_HIP(hipMalloc((void**)&d_counts, sizeof counts));
    CHECK_HIP(hipMalloc((void**)&d_norm, sizeof norm_dev));
    CHECK_HIP(hipMemcpy(d_counts, counts, sizeof counts, hipMemcpyHostToDevice));
    CHECK_HHSR(hhsr_normalize_raw_u16(d_counts, 1, H, W, W, cfa, black, white, wb, d_norm, NULL));
    CHECK_HIP(hipMemcpy(norm_dev, d_norm, sizeof norm_dev, hipMemcpyDeviceToHost));
    for (int i = 0; i < H * W; ++i)
        if (norm_dev[i] != norm_host[i]) { fprintf(stderr, "normalise mismatch at %d: %g vs %g\n", i, norm_dev[i], norm_host[i]); return 1; }

    /* one pyramid level: 9-tap Gaussian (sigma = 1), valid convolution, decimate by 2 */
    float taps[NT], sum = 0.f;
    for (int k = 0; k < NT; ++k) { taps[k] = expf(-0.5f * (float)((k - 4) * (k - 4))); sum += taps[k]; }
    for (int k = 0; k < NT; ++k) taps[k] /= sum;
    const int h2 = (H - 8) / F, w2 = (W - 8) / F;
    CHECK_HIP(hipMalloc((void**)&d_lvl, sizeof(float) * h2 * w2));
    CHECK_HHSR(hhsr_gauss_decimate(d_norm, H, W, W, d_lvl, w2, F, taps, NT, NULL));
    float* lvl = (float*)malloc(sizeof(float) * h2 * w2);
    CHECK_HIP(hipMemcpy(lvl, d_lvl, sizeof(float) * h2 * w2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int y = 0; y < h2; ++y)
        for (int x = 0; x < w2; ++x) {
            double acc = 0;
            for (int j = 0; j < NT; ++j) {
                double col = 0;
                for (int i = 0; i < NT; ++i) col += (double)taps[i] * norm_host[(y * F + i) * W + x * F + j];
                acc += (double)taps[j] * col;
            }
            const double d = fabs(acc - lvl[y * w2 + x]);
            if (d > worst) worst = d;
        }
    printf("normalise: bit-exact; pyramid level %dx%d: max abs diff vs float64 %.2e\n", h2, w2, worst);
    /* argument errors come back as codes + messages, never as exceptions */
    if (hhsr_gauss_decimate(d_norm, H, W, W, d_lvl, w2, 3, taps, NT, NULL) == 0) return 1;
    printf("rejected factor 3: %s\n", hhsr_last_error());
    free(lvl);
    hipFree(d_counts); hipFree(d_norm); hipFree(d_lvl);
    return worst < 1e-5 ? 0 : 1;
}

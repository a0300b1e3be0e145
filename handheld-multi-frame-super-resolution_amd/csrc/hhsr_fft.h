// Internal interface of the fused in-LDS FFT low-pass (hhsr_fft.hip), used by the grey plan (hhsr_grey.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define HHSR_MAX_RADICES 16

struct HhsrRadices {
    int n;
    int r[HHSR_MAX_RADICES];
    int pow_min;  // passes whose twiddle table would hold more than this many entries keep only w^k and raise it to the
                  // powers 2 .. R-1 in registers (hhsr_fft.hip: stockham_pass); <= 0: every pass has its full table
};

struct HhsrFft {
    bool ok = false;
    int H = 0, W = 0, Wk = 0;       // Wk: number of kept x-bins (kx = 0 .. Wk-1)
    HhsrRadices radM{}, radH{};     // radix schedules of the W/2-point and H-point transforms
    float2 *twM = nullptr, *twH = nullptr, *twW = nullptr;  // exp(-2 pi i k / {W/2, H, W})
    float2* T = nullptr;            // transposed kept half spectrum [Wk][H]
    size_t lds_rows = 0, lds_cols = 0;
    int twlenM = 0, twlenH = 0;     // lengths of the per-pass twiddle tables
    int rb = 0;                     // rows per workgroup of the row kernels (4, 2 or 1 by LDS budget)
    int nt_rows = 0;                // threads per workgroup of the row kernels (256 with rb = 1 when the row fits, else 512)
    int nc = 0;                     // kept columns per workgroup of the column kernel (2 or 1)
    int batch = 1;                  // frames one launch may carry: T holds this many spectra, tstride elements apart
    size_t tstride = 0;
    int static_rows = 0, static_cols = 0;  // > 0: the row / column kernels run the passes of this compile-time plan
                                           // (hhsr_fft.hip: HHSR_STATIC_ROWS / HHSR_STATIC_COLS)
};

bool hhsr_fft_create(HhsrFft& f, int H, int W, int batch);   // false: sizes unsupported (caller uses the library plans)
void hhsr_fft_destroy(HhsrFft& f);
// n frames: one launch per phase for every f.batch of them (row blocks of all frames walked by one resident round of workgroups)
int hhsr_fft_lowpass(const HhsrFft& f, const float* const* srcs, float* const* dsts, int n, hipStream_t s);

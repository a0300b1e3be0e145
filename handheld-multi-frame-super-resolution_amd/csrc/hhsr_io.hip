// Burst front end: sensor counts -> the normalised, white-balanced float32 RAW the hot path
// consumes (reference utils_dng.py:149-160, which does this on the host with NumPy after rawpy decoded the DNGs).
//
//   v = (float32(count) - black[c]) / (white - black[c]);  v *= wb[c] / wb[1]      c = CFA colour of the pixel
//
// in the reference's float32 arithmetic (float32 array op Python scalar -> float32): IEEE subtraction, division
// and multiplication, no contraction, so the result is bit-identical to the NumPy expression.  2 B in / 4 B out
// per pixel: HBM bound; each thread converts 8 pixels of one row (one 16-byte load, two 16-byte stores).
#include "hhsr_common.h"

struct NormArgs {
    float black[4], inv_unused, den[4], gain[4];  // per position of the 2x2 CFA cell: (row & 1) * 2 + (col & 1)
};

__global__ void __launch_bounds__(256) k_normalize_u16(const uint16_t* __restrict__ raw, int W, int pitch_in,
                                                        size_t frame_in, float* __restrict__ out, size_t frame_out,
                                                        int H, NormArgs A) {
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 8, y = blockIdx.y, n = blockIdx.z;
    if (x0 >= W) return;
    const uint16_t* __restrict__ src = raw + n * frame_in + (size_t)y * pitch_in + x0;
    float* __restrict__ dst = out + n * frame_out + (size_t)y * W + x0;
    const int r = (y & 1) * 2;
    const float b0 = A.black[r], b1 = A.black[r + 1], d0 = A.den[r], d1 = A.den[r + 1], g0 = A.gain[r],
                g1 = A.gain[r + 1];
    if (x0 + 8 <= W && ((pitch_in | W) & 7) == 0) {
        const uint4 p = *reinterpret_cast<const uint4*>(src);  // 8 counts; x0 is even: even lanes are CFA column 0
        const uint32_t w[4] = {p.x, p.y, p.z, p.w};
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[2 * k] = ((float)(w[k] & 0xffffu) - b0) / d0 * g0;
            v[2 * k + 1] = ((float)(w[k] >> 16) - b1) / d1 * g1;
        }
        reinterpret_cast<float4*>(dst)[0] = make_float4(v[0], v[1], v[2], v[3]);
        reinterpret_cast<float4*>(dst)[1] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        for (int k = 0; k < 8 && x0 + k < W; ++k) {
            const bool odd = k & 1;
            dst[k] = ((float)src[k] - (odd ? b1 : b0)) / (odd ? d1 : d0) * (odd ? g1 : g0);
        }
    }
}

extern "C" int hhsr_normalize_raw_u16(const uint16_t* raw, int n_frames, int H, int W, int pitch,
                                      const uint8_t cfa[4], const double* black_levels, double white_level,
                                      const double* white_balance, float* out, void* stream) {
    HHSR_ARG(raw && cfa && black_levels && white_balance && out);
    HHSR_ARG(n_frames > 0 && H > 0 && W > 0 && pitch >= W && n_frames <= 65535 && H <= 65535);
    HHSR_ARG(((uintptr_t)raw & 15) == 0 && ((uintptr_t)out & 15) == 0);
    HHSR_ARG(white_balance[1] != 0.0);
    NormArgs A;
    A.inv_unused = 0.f;
    for (int k = 0; k < 4; ++k) {
        HHSR_ARG(cfa[k] <= 2);
        const int c = cfa[k];
        HHSR_ARG(white_level != black_levels[c]);
        A.black[k] = (float)black_levels[c];                 // float32(array) - python scalar -> float32 scalar
        A.den[k] = (float)(white_level - black_levels[c]);   // python arithmetic first, then cast
        A.gain[k] = (float)(white_balance[c] / white_balance[1]);
    }
    const dim3 grid(hhsr_cdiv(hhsr_cdiv(W, 8), 256), H, n_frames);
    hipLaunchKernelGGL(k_normalize_u16, grid, dim3(256), 0, (hipStream_t)stream, raw, W, pitch, (size_t)H * pitch, out,
                       (size_t)H * W, H, A);
    HHSR_LAUNCHED();
}

// ---- shader-clock probe (measurement support: bench.py's "sclk_mhz") --------------------------------------------
// ONE wave reads the shader-cycle counter (s_memtime: one tick per shader clock, MI355X_MICROARCH.md "s_memtime tick")
// and the constant 100 MHz counter (s_memrealtime) when it starts, sleeps until `ticks` of the constant counter have
// passed (s_sleep: no issue slots taken from the kernels it runs next to) and reads both again.  Launched on a side
// stream around a timed region it reports the clock the other kernels ACTUALLY ran at: out = {memtime0, realtime0,
// memtime1, realtime1}; sclk = (out[2] - out[0]) / (out[3] - out[1]) x 100 MHz.
__global__ void __launch_bounds__(64) k_clock_probe(unsigned long long* __restrict__ out, long long ticks) {
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    unsigned long long r1 = r0;
    while ((long long)(r1 - r0) < ticks) {
        __builtin_amdgcn_s_sleep(127);
        r1 = wall_clock64();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    out[0] = c0;
    out[1] = r0;
    out[2] = c1;
    out[3] = r1;
}

extern "C" int hhsr_clock_probe(uint64_t* out4, int64_t ticks_100mhz, void* stream) {
    HHSR_ARG(out4 != nullptr);
    HHSR_ARG(ticks_100mhz >= 0 && ticks_100mhz <= 1000000000LL);  // at most 10 s
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out4,
                       (long long)ticks_100mhz);
    HHSR_LAUNCHED();
}

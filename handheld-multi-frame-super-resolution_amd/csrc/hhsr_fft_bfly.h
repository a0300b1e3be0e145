// Register butterflies of the in-LDS FFTs (hhsr_fft.hip; also included by the wave-synchronous experiment kept under
// tools/experiments/fft_wave/).
#pragma once
#include <hip/hip_runtime.h>

#define HHSR_FFT_POW_RMAX 6  // radices up to this may take the power form (the schedule puts the small radices last)
#ifndef HHSR_FFT_POW_MIN
#define HHSR_FFT_POW_MIN 1024  // entries of one pass's twiddle table above which it is kept as w^k only (0: never)
#endif

// ---- complex helpers ------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)

// ---- forward R-point DFTs (w = exp(-2 pi i / R)) ------------------------------------------------------------
__device__ __forceinline__ void dft2(float2* v) {
    const float2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
__device__ __forceinline__ void dft3(float2* v) {
    const float s = 0.86602540378443864676f;
    const float2 t1 = cadd(v[1], v[2]);
    const float2 m1 = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
    const float2 d = cscale(mul_mi(csub(v[1], v[2])), s);  // -i s (b - c)
    v[0] = cadd(v[0], t1);
    v[1] = cadd(m1, d);
    v[2] = csub(m1, d);
}
__device__ __forceinline__ void dft4(float2* v) {
    const float2 s02 = cadd(v[0], v[2]), d02 = csub(v[0], v[2]);
    const float2 s13 = cadd(v[1], v[3]), d13 = mul_mi(csub(v[1], v[3]));  // -i (b - d)
    v[0] = cadd(s02, s13);
    v[2] = csub(s02, s13);
    v[1] = cadd(d02, d13);
    v[3] = csub(d02, d13);
}
__device__ __forceinline__ void dft5(float2* v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const float2 t1 = cadd(v[1], v[4]), t2 = cadd(v[2], v[3]);
    const float2 t3 = csub(v[1], v[4]), t4 = csub(v[2], v[3]);
    const float2 a = v[0];
    const float2 m1 = make_float2(a.x + c1 * t1.x + c2 * t2.x, a.y + c1 * t1.y + c2 * t2.y);
    const float2 m2 = make_float2(a.x + c2 * t1.x + c1 * t2.x, a.y + c2 * t1.y + c1 * t2.y);
    const float2 n1 = mul_mi(make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y));  // -i n1
    const float2 n2 = mul_mi(make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y));  // -i n2
    v[0] = cadd(a, cadd(t1, t2));
    v[1] = cadd(m1, n1);
    v[4] = csub(m1, n1);
    v[2] = cadd(m2, n2);
    v[3] = csub(m2, n2);
}

// 7-point DFT: a_j = x_j + x_{7-j}, b_j = x_j - x_{7-j};  X_k = x_0 + sum_j a_j cos(2 pi j k / 7) -+ i sum_j b_j
// sin(2 pi j k / 7) for k and 7 - k.  (4032 x 3024 sensors: 2016 = 2^5 3^2 7, 3024 = 2^4 3^3 7.)
__device__ __forceinline__ void dft7(float2* v) {
    const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
    const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
    const float2 x0 = v[0];
    const float2 a1 = cadd(v[1], v[6]), a2 = cadd(v[2], v[5]), a3 = cadd(v[3], v[4]);
    const float2 b1 = csub(v[1], v[6]), b2 = csub(v[2], v[5]), b3 = csub(v[3], v[4]);
    // cos / sin of 2 pi j k / 7 for (j, k) in 1..3: index j*k mod 7 folded to 1..3 (sin changes sign past 3)
    const float2 m1 = make_float2(x0.x + c1 * a1.x + c2 * a2.x + c3 * a3.x, x0.y + c1 * a1.y + c2 * a2.y + c3 * a3.y);
    const float2 m2 = make_float2(x0.x + c2 * a1.x + c3 * a2.x + c1 * a3.x, x0.y + c2 * a1.y + c3 * a2.y + c1 * a3.y);
    const float2 m3 = make_float2(x0.x + c3 * a1.x + c1 * a2.x + c2 * a3.x, x0.y + c3 * a1.y + c1 * a2.y + c2 * a3.y);
    const float2 n1 = mul_mi(make_float2(s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y));
    const float2 n2 = mul_mi(make_float2(s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y));
    const float2 n3 = mul_mi(make_float2(s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y));
    v[0] = cadd(x0, cadd(a1, cadd(a2, a3)));
    v[1] = cadd(m1, n1);
    v[6] = csub(m1, n1);
    v[2] = cadd(m2, n2);
    v[5] = csub(m2, n2);
    v[3] = cadd(m3, n3);
    v[4] = csub(m3, n3);
}

// ---- composite radices: R = A * B point DFTs in registers ----------------------------------------------------
// Cooley-Tukey inside the butterfly: n = n1 B + n2, k = k1 + A k2:
//   X[k1 + A k2] = sum_n2 w_B^(n2 k2) [ w_R^(n2 k1) sum_n1 v[n1 B + n2] w_A^(n1 k1) ].
// The inner twiddles w_R^m are compile-time constants (float64 Taylor series, constexpr), trivial ones
// (1, -i, -1, +i) cost no multiplication.  Three passes of radix 10-25 replace the five to six passes of radix
// <= 5: half the LDS round trips and workgroup barriers of kernels that are bound by exactly those.
constexpr double hhsr_pi = 3.14159265358979323846264338327950288;
constexpr double c_sin_small(double x) {  // |x| <= pi/4
    double term = x, sum = x;
    for (int n = 1; n < 12; ++n) {
        term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double c_cos_small(double x) {
    double term = 1.0, sum = 1.0;
    for (int n = 1; n < 12; ++n) {
        term *= -x * x / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}
// cos / sin of 2 pi m / R through octant reduction (exact on the axes)
constexpr double c_cos2pi(int m, int R) {
    m %= R;
    if (8 * m <= R) return c_cos_small(2.0 * hhsr_pi * m / R);
    if (8 * m <= 3 * R) return -c_sin_small(2.0 * hhsr_pi * (4 * m - R) / (4.0 * R));       // cos(pi/2 + d) = -sin d
    if (8 * m <= 5 * R) return -c_cos_small(2.0 * hhsr_pi * (2 * m - R) / (2.0 * R));       // cos(pi + d) = -cos d
    if (8 * m <= 7 * R) return c_sin_small(2.0 * hhsr_pi * (4 * m - 3 * R) / (4.0 * R));    // cos(3pi/2 + d) = sin d
    return c_cos_small(2.0 * hhsr_pi * (m - R) / (double)R);
}
constexpr double c_sin2pi(int m, int R) {
    m %= R;
    if (8 * m <= R) return c_sin_small(2.0 * hhsr_pi * m / R);
    if (8 * m <= 3 * R) return c_cos_small(2.0 * hhsr_pi * (4 * m - R) / (4.0 * R));
    if (8 * m <= 5 * R) return -c_sin_small(2.0 * hhsr_pi * (2 * m - R) / (2.0 * R));
    if (8 * m <= 7 * R) return -c_cos_small(2.0 * hhsr_pi * (4 * m - 3 * R) / (4.0 * R));
    return c_sin_small(2.0 * hhsr_pi * (m - R) / (double)R);
}
template <int R>
struct TwTab {  // w_R^m = exp(-2 pi i m / R)
    float c[R], s[R];
    constexpr TwTab() : c(), s() {
        for (int m = 0; m < R; ++m) {
            c[m] = (float)c_cos2pi(m, R);
            s[m] = (float)(-c_sin2pi(m, R));
        }
    }
};

template <int R>
__device__ __forceinline__ void dft_reg(float2* v);
template <> __device__ __forceinline__ void dft_reg<2>(float2* v) { dft2(v); }
template <> __device__ __forceinline__ void dft_reg<3>(float2* v) { dft3(v); }
template <> __device__ __forceinline__ void dft_reg<4>(float2* v) { dft4(v); }
template <> __device__ __forceinline__ void dft_reg<5>(float2* v) { dft5(v); }
template <> __device__ __forceinline__ void dft_reg<7>(float2* v) { dft7(v); }

template <int A, int B>
__device__ __forceinline__ void dft_comp(float2* v) {
    constexpr int R = A * B;
    constexpr TwTab<R> tab{};
    float2 y[B][A];
#pragma unroll
    for (int n2 = 0; n2 < B; ++n2) {
        float2 t[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) t[n1] = v[n1 * B + n2];
        dft_reg<A>(t);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) {
            const int m = (n2 * k1) % R;
            if (m == 0) y[n2][k1] = t[k1];
            else if (4 * m == R) y[n2][k1] = mul_mi(t[k1]);
            else if (2 * m == R) y[n2][k1] = make_float2(-t[k1].x, -t[k1].y);
            else if (4 * m == 3 * R) y[n2][k1] = mul_pi(t[k1]);
            else y[n2][k1] = cmul(t[k1], make_float2(tab.c[m], tab.s[m]));
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < A; ++k1) {
        float2 t[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) t[n2] = y[n2][k1];
        dft_reg<B>(t);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) v[k1 + A * k2] = t[k2];
    }
}
template <> __device__ __forceinline__ void dft_reg<6>(float2* v) { dft_comp<3, 2>(v); }
template <> __device__ __forceinline__ void dft_reg<8>(float2* v) { dft_comp<4, 2>(v); }
template <> __device__ __forceinline__ void dft_reg<9>(float2* v) { dft_comp<3, 3>(v); }
template <> __device__ __forceinline__ void dft_reg<10>(float2* v) { dft_comp<5, 2>(v); }
template <> __device__ __forceinline__ void dft_reg<12>(float2* v) { dft_comp<4, 3>(v); }
template <> __device__ __forceinline__ void dft_reg<14>(float2* v) { dft_comp<7, 2>(v); }
template <> __device__ __forceinline__ void dft_reg<15>(float2* v) { dft_comp<5, 3>(v); }
template <> __device__ __forceinline__ void dft_reg<16>(float2* v) { dft_comp<4, 4>(v); }


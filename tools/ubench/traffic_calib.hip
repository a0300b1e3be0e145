// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of k_merge_x2.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/traffic_calib tools/ubench/traffic_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/tc_f -o f -- tools/ubench/traffic_calib
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/tc_w -o w -- tools/ubench/traffic_calib
//   python tools/ubench/traffic_calib_report.py /tmp/tc_f /tmp/tc_w
// Every kernel moves a KNOWN number of bytes over buffers far larger than the 256 MB Infinity Cache (each launch touches
// 2.4 GB), each read kernel reads every byte exactly once per launch (except `window`, whose requested / unique ratio is
// the pattern under test), each write kernel writes every byte exactly once.  The program prints the true byte counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int W = 24576, H = 24576;             // float plane: 2.4 GB
constexpr size_t N = (size_t)W * H;

// (1) wide coalesced streaming read: 16 B per lane (the guide's calibrated case)
__global__ void __launch_bounds__(256) read_b128(const float4* __restrict__ p, float* __restrict__ sink, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) sink[0] = acc;
}
// (2) coalesced dword read: 4 B per lane
__global__ void __launch_bounds__(256) read_b32(const float* __restrict__ p, float* __restrict__ sink, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 123.456f) sink[0] = acc;
}
// (3) k_merge_x2's staging: one workgroup per 16 x 16 tile reads the 19 x 19 window around it with dword loads (361
// loads by 256 threads: 105 threads load twice), tiles walked in raster order — requested bytes = 361 / 256 x the plane,
// unique bytes = the plane (+ a 3-pixel frame); neighbouring windows overlap by 3 columns / rows
__global__ void __launch_bounds__(256) read_window(const float* __restrict__ p, float* __restrict__ sink) {
    const int tx0 = blockIdx.x * 16, ty0 = blockIdx.y * 16;
    float acc = 0.f;
    for (int e = threadIdx.x; e < 19 * 19; e += 256) {
        const int ey = e / 19, ex = e - ey * 19;
        const int y = min(ty0 + ey, H - 1), x = min(tx0 + ex, W - 1);
        acc += p[(size_t)y * W + x];
    }
    if (acc == 123.456f) sink[0] = acc;
}
// (4) wide coalesced streaming write: 16 B per lane
__global__ void __launch_bounds__(256) write_b128(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// (5) coalesced dword write
__global__ void __launch_bounds__(256) write_b32(float* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 1.f;
}
// (6) k_merge_x2's tile store: one workgroup writes a 32-row x 384-byte tile (96 floats per row as 24 float4: 768 float4
// by 256 threads in 3 rounds) into an image whose rows are W floats long — 384-byte row segments at a 98 304-byte pitch
__global__ void __launch_bounds__(256) write_tile384(float* __restrict__ p) {
    const size_t x0 = (size_t)blockIdx.x * 96, y0 = (size_t)blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int q = threadIdx.x + 256 * r, row = q / 24, c = (q - row * 24) * 4;
        *reinterpret_cast<float4*>(p + (y0 + row) * W + x0 + c) = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}
// (7) the first-generation store of the same tile: 3 dwords per thread and output pixel (12-byte records)
__global__ void __launch_bounds__(256) write_tile_dwords(float* __restrict__ p) {
    const size_t x0 = (size_t)blockIdx.x * 96, y0 = (size_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int sa = 0; sa < 2; ++sa)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int k = 0; k < 3; ++k) p[(y0 + 2 * ty + sa) * W + x0 + (2 * tx + sb) * 3 + k] = 1.f;
}

int main() {
    float *a, *sink;
    CHECK(hipMalloc((void**)&a, N * sizeof(float)));
    CHECK(hipMalloc((void**)&sink, 256));
    CHECK(hipMemset(a, 0, N * sizeof(float)));
    const dim3 b(256);
    const int g = 256 * 32;
    const double bytes = (double)N * 4;
    for (int rep = 0; rep < 2; ++rep) {  // (the report averages the dispatches of each kernel)
        hipLaunchKernelGGL(read_b128, dim3(g), b, 0, 0, (const float4*)a, sink, N / 4);
        hipLaunchKernelGGL(read_b32, dim3(g), b, 0, 0, a, sink, N);
        hipLaunchKernelGGL(read_window, dim3(W / 16, H / 16), b, 0, 0, a, sink);
        hipLaunchKernelGGL(write_b128, dim3(g), b, 0, 0, (float4*)a, N / 4);
        hipLaunchKernelGGL(write_b32, dim3(g), b, 0, 0, a, N);
        hipLaunchKernelGGL(write_tile384, dim3(W / 96, H / 32), b, 0, 0, a);
        hipLaunchKernelGGL(write_tile_dwords, dim3(W / 96, H / 32), b, 0, 0, a);
        CHECK(hipDeviceSynchronize());
    }
    printf("TRUE read_b128 %.0f\nTRUE read_b32 %.0f\nTRUE read_window %.0f requested %.0f\nTRUE write_b128 %.0f\n"
           "TRUE write_b32 %.0f\nTRUE write_tile384 %.0f\nTRUE write_tile_dwords %.0f\n",
           bytes, bytes, bytes, bytes * 361.0 / 256.0, bytes, bytes, bytes, bytes);
    return 0;
}

// Per-level alignment kernels: Lucas-Kanade precompute, block matching, ICA, flow upscaling
// (reference ICA.py, block_matching.py, alignment.py:150-172).
//
// One 256-thread workgroup (4 wave64) per tile.  The moving search window and the reference tile are
// staged in LDS once and reused by all (2r+1)^2 candidate shifts; per-tile sums use wave64 shuffles
// plus a 4-entry LDS exchange.  These levels are small (sum of level sizes = 1.33 P) and latency bound,
// not MFMA shaped: 81 shifts x ts^2 MACs per tile is ~0.1 GFLOP for a 12 MP level.
#include "hhsr_common.h"

// ---- gradients + per-tile Hessian (ICA.py:15-76) ------------------------------------------------
__global__ void __launch_bounds__(256) k_grad_hessian(const float* __restrict__ I, int H, int W, int pitch, int ts,
                                                       float* __restrict__ gx, float* __restrict__ gy,
                                                       float* __restrict__ hess, int ny, int nx) {
    __shared__ float sm[12];
    const int tx = blockIdx.x, ty = blockIdx.y;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int p = threadIdx.x; p < ts * ts; p += 256) {
        const int i = p / ts, j = p - i * ts;
        const int y = ty * ts + i, x = tx * ts + j;
        if (y < H && x < W) {
            const float l = x > 0 ? I[(size_t)y * pitch + x - 1] : 0.f;
            const float r = x + 1 < W ? I[(size_t)y * pitch + x + 1] : 0.f;
            const float u = y > 0 ? I[(size_t)(y - 1) * pitch + x] : 0.f;
            const float d = y + 1 < H ? I[(size_t)(y + 1) * pitch + x] : 0.f;
            const float vx = r - l, vy = d - u;
            gx[(size_t)y * pitch + x] = vx;
            gy[(size_t)y * pitch + x] = vy;
            a += vx * vx;
            b += vx * vy;
            c += vy * vy;
        }
    }
    float c2 = 0.f;
    block_sum2<4>(a, b, sm);
    block_sum2<4>(c, c2, sm);
    if (threadIdx.x == 0 && ty < ny && tx < nx) {
        float* h = hess + ((size_t)ty * nx + tx) * 4;
        h[0] = a;
        h[1] = b;
        h[2] = b;
        h[3] = c;
    }
}

extern "C" int hhsr_grad_hessian(const float* lvl, int H, int W, int pitch, int ts, float* gx, float* gy,
                                 float* hess, void* stream) {
    HHSR_ARG(lvl && gx && gy && hess && H > 0 && W > 0 && pitch >= W);
    HHSR_ARG(ts == 8 || ts == 16 || ts == 32 || ts == 64);
    const int ny = H / ts, nx = W / ts;
    hipLaunchKernelGGL(k_grad_hessian, dim3(hhsr_cdiv(W, ts), hhsr_cdiv(H, ts)), dim3(256), 0, (hipStream_t)stream,
                       lvl, H, W, pitch, ts, gx, gy, hess, ny, nx);
    HHSR_LAUNCHED();
}

// ---- block matching -----------------------------------------------------------------------------
// LDS layout: s_ref[ts*ts] | s_win[P*Pp] (P = ts+2r, Pp = P|1) | s_part[256] | s_cost[n] ...
// Thread t handles candidate c = t % n for the pixel subset g = t / n (g < G = 256 / n), so that the
// lanes of a wave read consecutive window addresses (conflict-free) and one broadcast ref address.
template <bool L1>
__global__ void __launch_bounds__(256) k_block_match(const float* __restrict__ ref, int ref_pitch,
                                                      const float* __restrict__ mov, int mh, int mw, int mov_pitch,
                                                      float* __restrict__ flow, int nx, int ts, int r, int mode) {
    extern __shared__ float lds[];
    const int tx = blockIdx.x, ty = blockIdx.y;
    float* fl = flow + ((size_t)ty * nx + tx) * 2;
    const float f0 = fl[0], f1 = fl[1];
    const float r0 = rintf(f0), r1 = rintf(f1);  // round-half-even (torch.round / Python round)
    if (L1 && mode == 1) {                        // "L1_ref_effective": flow <- round(flow)
        if (threadIdx.x == 0) {
            fl[0] = r0;
            fl[1] = r1;
        }
        return;
    }
    const int P = ts + 2 * r, Pp = P | 1, n1 = 2 * r + 1, n = n1 * n1;
    float* s_ref = lds;
    float* s_win = s_ref + ts * ts;
    float* s_part = s_win + P * Pp;
    const int tid = threadIdx.x;
    const int y0 = ty * ts + (int)r1 - r, x0 = tx * ts + (int)r0 - r;
    for (int p = tid; p < ts * ts; p += 256) {
        const int i = p / ts, j = p - i * ts;
        s_ref[p] = ref[(size_t)(ty * ts + i) * ref_pitch + tx * ts + j];
    }
    for (int p = tid; p < P * P; p += 256) {
        const int i = p / P, j = p - i * P;
        const int y = y0 + i, x = x0 + j;
        float v;
        if (L1) {  // zero outside the moving level (block_matching.py:131-139)
            v = (y >= 0 && y < mh && x >= 0 && x < mw) ? mov[(size_t)y * mov_pitch + x] : 0.f;
        } else {   // clamp-to-edge (block_matching.py:369-371)
            v = mov[(size_t)clampi(y, 0, mh - 1) * mov_pitch + clampi(x, 0, mw - 1)];
        }
        s_win[i * Pp + j] = v;
    }
    __syncthreads();
    const int G = max(1, min(256 / n, 16));
    float best = INFINITY;
    int besti = 0;
    // candidates in rounds of `n_per_round` (n can exceed 256 for large radii)
    for (int cbase = 0; cbase < n; cbase += 256) {
        const int nc = min(n - cbase, 256);
        const int Gr = (nc == n) ? G : 1;
        const int c = cbase + tid % nc, g = tid / nc;
        float acc = 0.f;
        if (g < Gr) {
            const int dy = c / n1, dx = c - dy * n1;
            for (int i = g; i < ts; i += Gr) {
                const float* wrow = s_win + (i + dy) * Pp + dx;
                const float* rrow = s_ref + i * ts;
                for (int j = 0; j < ts; ++j) {
                    const float d = rrow[j] - wrow[j];
                    acc += L1 ? fabsf(d) : d * d;
                }
            }
        }
        __syncthreads();
        s_part[tid] = acc;
        __syncthreads();
        // candidate totals (fixed order over g), then this thread keeps a running first-minimum
        if (tid < nc) {
            float tot = s_part[tid];
            for (int k = 1; k < Gr; ++k) tot += s_part[k * nc + tid];
            if (tot < best) {  // strict: earlier candidate wins ties within this thread's sequence
                best = tot;
                besti = cbase + tid;
            }
        }
    }
    // first minimum in row-major candidate order == torch.argmin on the flattened map
    __syncthreads();
    float* s_cost = s_part;
    int* s_idx = reinterpret_cast<int*>(s_part + 256);
    s_cost[tid] = best;
    s_idx[tid] = besti;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float oc = s_cost[tid + s];
            const int oi = s_idx[tid + s];
            if (oc < s_cost[tid] || (oc == s_cost[tid] && oi < s_idx[tid])) {
                s_cost[tid] = oc;
                s_idx[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int bi = s_idx[0];
        const int dy = bi / n1 - r, dx = bi % n1 - r;
        if (L1) {  // flow <- round(flow) + shift (block_matching.py:119-120,179-180)
            fl[0] = r0 + (float)dx;
            fl[1] = r1 + (float)dy;
        } else {   // shift added to the UN-rounded flow (block_matching.py:75-76)
            fl[0] = f0 + (float)dx;
            fl[1] = f1 + (float)dy;
        }
    }
}

// ---- block matching, one wave64 per tile ------------------------------------------------------------
// 4 tiles per 256-thread workgroup; each wave stages its ts^2 reference tile and (ts+2r)^2 moving window
// in its own LDS slice.  Two work mappings, chosen by the number of candidates n = (2r+1)^2:
//   n <= 16 (level 0, r = 1): lane = pixel subset; per candidate a 6-step xor-butterfly sum;
//   n  > 16 (r = 4: 81):      lane = candidate (ceil(n/64) rounds), each lane walks the whole tile reading
//                             one broadcast reference word and consecutive window words per step;
// then a (cost, index)-lexicographic butterfly argmin = first minimum in row-major order.
// One pixel of an ICA iteration (ICA.py:160-180 / 246-266): bilinear sample of the moving level, temporal
// gradient, accumulation of B = -sum grad * gradt.  Written with fused multiply-adds (7 instructions instead of 12):
// the level-0 launch is VALU-issue bound.  NVVM contracts the same expressions in the reference's kernels on its
// own hardware; against the un-fused NumPy restatement the flows move by ~1e-6 px.
__device__ __forceinline__ void ica_tap(float m00, float m01, float m10, float m11, float frx, float fry, float rcv,
                                        float gx, float gy, float& B0, float& B1) {
    const float top = fmaf(m01 - m00, frx, m00);
    const float bot = fmaf(m11 - m10, frx, m10);
    const float gradt = fmaf(bot - top, fry, top) - rcv;
    B0 = fmaf(-gx, gradt, B0);
    B1 = fmaf(-gy, gradt, B1);
}

struct CostIdx {
    float c;
    int i;
};

__device__ __forceinline__ CostIdx wave_argmin(CostIdx v) {
    CostIdx o;
    o.i = wave_argmin_first(v.c, v.i, &o.c);
    return o;
}

template <int TS, bool L1>
__global__ void __launch_bounds__(256) k_bm_wave(const float* __restrict__ ref, int ref_pitch,
                                                  const float* __restrict__ mov, int mh, int mw, int mov_pitch,
                                                  float* __restrict__ flow, int nx, int ntiles, int r, int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int ts = TS;
    const int wave = threadIdx.x / HHSR_WAVE, lane = threadIdx.x & (HHSR_WAVE - 1);
    const int P = ts + 2 * r, Pp = P | 1, n1 = 2 * r + 1, n = n1 * n1;
    const int nparts = 4 * n;                       // (candidate, row quarter) work items, n > 16 path
    const int slice = (ts * ts + P * Pp + nparts + 3) & ~3;
    float* s_ref = lds + (size_t)wave * slice;
    float* s_win = s_ref + ts * ts;
    float* s_part = s_win + P * Pp;
    const int tile = blockIdx.x * 4 + wave;
    bool active = tile < ntiles;
    const int tsafe = active ? tile : 0;
    const int ty = tsafe / nx, tx = tsafe - ty * nx;
    float* fl = flow + (size_t)tsafe * 2;
    const float f0 = fl[0], f1 = fl[1];
    const float r0 = rintf(f0), r1 = rintf(f1);  // round-half-even (torch.round / Python round)
    if (L1 && mode == 1) {                        // "L1_ref_effective": flow <- round(flow)
        if (active && lane == 0) {
            fl[0] = r0;
            fl[1] = r1;
        }
        return;  // block-uniform (mode is a kernel argument)
    }
    if (active) {
        const int y0 = ty * ts + (int)r1 - r, x0 = tx * ts + (int)r0 - r;
        for (int p = lane; p < ts * ts; p += HHSR_WAVE) {
            const int i = p / ts, j = p % ts;
            s_ref[p] = ref[(size_t)(ty * ts + i) * ref_pitch + tx * ts + j];
        }
        const float rcpP = 1.0f / (float)P;
        for (int p = lane; p < P * P; p += HHSR_WAVE) {
            const int i = (int)(((float)p + 0.5f) * rcpP), j = p - i * P;  // exact floor(p / P) for p < 2^16
            const int y = y0 + i, x = x0 + j;
            float v;
            if (L1) {  // zero outside the moving level (block_matching.py:131-139)
                v = (y >= 0 && y < mh && x >= 0 && x < mw) ? mov[(size_t)y * mov_pitch + x] : 0.f;
            } else {   // clamp-to-edge (block_matching.py:369-371)
                v = mov[(size_t)clampi(y, 0, mh - 1) * mov_pitch + clampi(x, 0, mw - 1)];
            }
            s_win[i * Pp + j] = v;
        }
    }
    __syncthreads();
    CostIdx best{INFINITY, 0};
    if (n <= 16) {
        if (!active) return;
        float rv[TS * TS / HHSR_WAVE > 0 ? TS * TS / HHSR_WAVE : 1];
        constexpr int PPT = TS * TS / HHSR_WAVE;
#pragma unroll
        for (int k = 0; k < PPT; ++k) rv[k] = s_ref[lane + k * HHSR_WAVE];
        for (int c = 0; c < n; ++c) {
            const int dy = c / n1, dx = c - dy * n1;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int p = lane + k * HHSR_WAVE;
                const float d = rv[k] - s_win[(p / ts + dy) * Pp + p % ts + dx];
                acc += L1 ? fabsf(d) : d * d;
            }
            acc = wave_sum_uniform(acc);
            if (acc < best.c) {  // strict: the first minimum wins
                best.c = acc;
                best.i = c;
            }
        }
    } else {
        // work item = (candidate, quarter of the tile rows); 4n items over 64 lanes
        if (active) {
            constexpr int QR = TS / 4;  // rows per quarter
            for (int it = lane; it < nparts; it += HHSR_WAVE) {
                const int c = it >> 2, q = it & 3;
                const int dy = c / n1, dx = c - dy * n1;
                float acc = 0.f;
                for (int i = q * QR; i < (q + 1) * QR; ++i) {
                    const float* wrow = s_win + (i + dy) * Pp + dx;
                    const float4* rrow = reinterpret_cast<const float4*>(s_ref + i * ts);
#pragma unroll
                    for (int j4 = 0; j4 < TS / 4; ++j4) {
                        const float4 rr = rrow[j4];  // one broadcast 16-byte LDS read
                        const float d0 = rr.x - wrow[4 * j4], d1 = rr.y - wrow[4 * j4 + 1];
                        const float d2 = rr.z - wrow[4 * j4 + 2], d3 = rr.w - wrow[4 * j4 + 3];
                        if (L1) acc += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
                        else acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
                    }
                }
                s_part[it] = acc;
            }
        }
        __syncthreads();
        if (!active) return;
        for (int c = lane; c < n; c += HHSR_WAVE) {
            const float tot = (s_part[4 * c] + s_part[4 * c + 1]) + (s_part[4 * c + 2] + s_part[4 * c + 3]);
            if (tot < best.c) {
                best.c = tot;
                best.i = c;
            }
        }
        best = wave_argmin(best);
    }
    if (lane == 0) {
        const int dy = best.i / n1 - r, dx = best.i % n1 - r;
        if (L1) {  // flow <- round(flow) + shift (block_matching.py:119-120,179-180)
            fl[0] = r0 + (float)dx;
            fl[1] = r1 + (float)dy;
        } else {   // shift added to the UN-rounded flow (block_matching.py:75-76)
            fl[0] = f0 + (float)dx;
            fl[1] = f1 + (float)dy;
        }
    }
}

static size_t bm_lds(int ts, int r) {
    const int P = ts + 2 * r, Pp = P | 1;
    return (size_t)(ts * ts + P * Pp + 512) * sizeof(float);
}

static size_t bm_wave_lds(int ts, int r) {
    const int P = ts + 2 * r, Pp = P | 1, n = (2 * r + 1) * (2 * r + 1);
    return (size_t)4 * ((ts * ts + P * Pp + 4 * n + 3) & ~3) * sizeof(float);
}

template <bool L1>
static void bm_launch(const float* ref, int ref_pitch, const float* mov, int mh, int mw, int mov_pitch, float* flow,
                      int ny, int nx, int ts, int r, int mode, hipStream_t s) {
    if (ts <= 32 && bm_wave_lds(ts, r) <= 56 * 1024) {
        const int ntiles = nx * ny;
        const dim3 g(hhsr_cdiv(ntiles, 4)), b(256);
        const size_t l = bm_wave_lds(ts, r);
#define BMW(TS) hipLaunchKernelGGL((k_bm_wave<TS, L1>), g, b, l, s, ref, ref_pitch, mov, mh, mw, mov_pitch, flow, nx, \
                                   ntiles, r, mode)
        if (ts == 8) BMW(8);
        else if (ts == 16) BMW(16);
        else BMW(32);
#undef BMW
    } else {  // large tiles (ts = 64): one workgroup per tile
        hipLaunchKernelGGL(k_block_match<L1>, dim3(nx, ny), dim3(256), bm_lds(ts, r), s, ref, ref_pitch, mov, mh, mw,
                           mov_pitch, flow, nx, ts, r, mode);
    }
}

extern "C" int hhsr_bm_l2(const float* ref, int ref_pitch, const float* mov, int mh, int mw, int mov_pitch,
                          float* flow, int ny, int nx, int ts, int r, void* stream) {
    HHSR_ARG(ref && mov && flow && mh > 0 && mw > 0 && ny > 0 && nx > 0 && r >= 0);
    HHSR_ARG(ts == 8 || ts == 16 || ts == 32 || ts == 64);  // the reference's box filters (block_matching.py:47-57)
    HHSR_ARG(bm_lds(ts, r) <= 64 * 1024);
    bm_launch<false>(ref, ref_pitch, mov, mh, mw, mov_pitch, flow, ny, nx, ts, r, 0, (hipStream_t)stream);
    HHSR_LAUNCHED();
}

extern "C" int hhsr_bm_l1(const float* ref, int ref_pitch, const float* mov, int mh, int mw, int mov_pitch,
                          float* flow, int ny, int nx, int ts, int r, int mode, void* stream) {
    HHSR_ARG(ref && mov && flow && mh > 0 && mw > 0 && ny > 0 && nx > 0 && r >= 0);
    HHSR_ARG(ts == 16 || ts == 32 || ts == 64);  // ts = 8 raises NotImplementedError upstream (block_matching.py:87)
    HHSR_ARG(mode == 0 || mode == 1);
    HHSR_ARG(bm_lds(ts, r) <= 64 * 1024);
    bm_launch<true>(ref, ref_pitch, mov, mh, mw, mov_pitch, flow, ny, nx, ts, r, mode, (hipStream_t)stream);
    HHSR_LAUNCHED();
}

// ---- ICA ------------------------------------------------------------------------------------------
// TS*TS/NT pixels per thread; reference value and both gradients stay in registers across iterations.
template <int TS, int NT>
__global__ void __launch_bounds__(NT) k_ica(const float* __restrict__ ref, const float* __restrict__ gx,
                                            const float* __restrict__ gy, int ref_pitch,
                                            const float* __restrict__ hess, const float* __restrict__ mov, int mh,
                                            int mw, int mov_pitch, float* __restrict__ flow, int nx, int n_iter,
                                            int row_bug) {
    constexpr int PPT = TS * TS / NT;  // pixels per thread
    constexpr int NW = NT / HHSR_WAVE;
    __shared__ float sm[2 * (NW > 0 ? NW : 1)];
    __shared__ float s_flow[2];
    const int tx = blockIdx.x, ty = blockIdx.y, tid = threadIdx.x;
    const float* h = hess + ((size_t)ty * nx + tx) * 4;
    const float A00 = h[0], A01 = h[1], A10 = h[2], A11 = h[3];
    const float det = A00 * A11 - A01 * A10;
    if (fabsf(det) < 1e-10f) return;  // not solvable: tile untouched (ICA.py:124-125)
    const float det_inv = 1.0f / det;
    float* fl = flow + ((size_t)ty * nx + tx) * 2;
    if (tid == 0) {
        s_flow[0] = fl[0];
        s_flow[1] = fl[1];
    }
    float rc[PPT], lgx[PPT], lgy[PPT];
    int py[PPT], px[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * NT;
        const int i = p / TS, j = p % TS;
        py[k] = ty * TS + i;
        px[k] = tx * TS + j;
        const size_t o = (size_t)py[k] * ref_pitch + px[k];
        rc[k] = ref[o];
        lgx[k] = gx[o];
        lgy[k] = gy[o];
    }
    for (int it = 0; it < n_iter; ++it) {
        __syncthreads();
        const float fxv = s_flow[0], fyv = s_flow[1];
        const float tx_ = truncf(fxv), ty_ = truncf(fyv);
        const float frx = fxv - tx_, fry = fyv - ty_;  // signed fraction of modf (D11)
        const int ix = (int)tx_, iy = (int)ty_;
        float B0 = 0.f, B1 = 0.f;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            int x0 = px[k] + ix, y0 = py[k] + iy;
            float m00, m01, m10, m11;
            if (TS == 8) {  // clamp-to-edge sampling (ICA.py:152-156)
                x0 = clampi(x0, 0, mw - 1);
                y0 = clampi(y0, 0, mh - 1);
                const int x1 = clampi(x0 + 1, 0, mw - 1), y1 = clampi(y0 + 1, 0, mh - 1);
                m00 = mov[(size_t)y0 * mov_pitch + x0];
                m01 = mov[(size_t)y0 * mov_pitch + x1];
                m10 = mov[(size_t)y1 * mov_pitch + x0];
                m11 = mov[(size_t)y1 * mov_pitch + x1];
            } else {  // zero outside (ICA.py:240-243)
                int yt = y0, yb = y0 + 1;
                if (TS == 64 && row_bug) {  // ICA.py:437-449 (D2): thread rows (4q .. 4q+3)
                    const int q = (py[k] - ty * TS) & 3;
                    yt = q == 0 ? y0 : y0 + 1;
                    yb = y0 + 2;
                }
                const bool xa = x0 >= 0 && x0 < mw, xb = x0 + 1 >= 0 && x0 + 1 < mw;
                const bool ya = yt >= 0 && yt < mh, yb_ = yb >= 0 && yb < mh;
                m00 = (ya && xa) ? mov[(size_t)yt * mov_pitch + x0] : 0.f;
                m01 = (ya && xb) ? mov[(size_t)yt * mov_pitch + x0 + 1] : 0.f;
                m10 = (yb_ && xa) ? mov[(size_t)yb * mov_pitch + x0] : 0.f;
                m11 = (yb_ && xb) ? mov[(size_t)yb * mov_pitch + x0 + 1] : 0.f;
            }
            ica_tap(m00, m01, m10, m11, frx, fry, rc[k], lgx[k], lgy[k], B0, B1);
        }
        block_sum2<NW>(B0, B1, sm);
        if (tid == 0) {
            s_flow[0] = fxv + det_inv * (A11 * B0 - A01 * B1);
            s_flow[1] = fyv + det_inv * (-A10 * B0 + A00 * B1);
        }
    }
    __syncthreads();
    if (tid == 0) {
        fl[0] = s_flow[0];
        fl[1] = s_flow[1];
    }
}

// ---- ICA, one wave64 per tile (TS = 16 / 32) ------------------------------------------------------------
// 4 tiles per 256-thread workgroup, no workgroup-wide reductions: each wave stages the (TS + 2M + 1)^2
// moving window around trunc(flow_0) in its own LDS slice once (zero outside the level, ICA.py:240-243),
// takes the 4 bilinear taps of every iteration from LDS, and sums (B0, B1) with a 6-step xor butterfly.
// If an update ever moves trunc(flow) more than M pixels from trunc(flow_0) (wave-uniform test) that
// iteration samples global memory instead.  ~4.7 vector loads per pixel instead of 15.
constexpr int ICA_M = 2;

__device__ __forceinline__ float wave_allsum(float v) { return wave_sum_uniform(v); }

template <int TS>
__global__ void __launch_bounds__(256) k_ica_wave(const float* __restrict__ ref, const float* __restrict__ gx,
                                                   const float* __restrict__ gy, int ref_pitch,
                                                   const float* __restrict__ hess, const float* __restrict__ mov,
                                                   int mh, int mw, int mov_pitch, float* __restrict__ flow, int nx,
                                                   int ntiles, int n_iter) {
    constexpr int PPT = TS * TS / HHSR_WAVE;
    constexpr int WS = TS + 2 * ICA_M + 1, WP = WS | 1;
    __shared__ float s_win[4][WS * WP];
    const int wave = threadIdx.x / HHSR_WAVE, lane = threadIdx.x & (HHSR_WAVE - 1);
    const int tile = blockIdx.x * 4 + wave;
    bool active = tile < ntiles;
    const int tsafe = active ? tile : 0;
    const int ty = tsafe / nx, tx = tsafe - ty * nx;
    const float* h = hess + (size_t)tsafe * 4;
    const float A00 = h[0], A01 = h[1], A10 = h[2], A11 = h[3];
    const float det = A00 * A11 - A01 * A10;
    active = active && !(fabsf(det) < 1e-10f);  // not solvable: tile untouched (ICA.py:212-213)
    const float det_inv = 1.0f / det;
    float fxv = flow[(size_t)tsafe * 2], fyv = flow[(size_t)tsafe * 2 + 1];
    const int ix0 = (int)truncf(fxv), iy0 = (int)truncf(fyv);
    const int ox = tx * TS + ix0 - ICA_M, oy = ty * TS + iy0 - ICA_M;
    float* win = s_win[wave];
    if (active) {
        for (int p = lane; p < WS * WS; p += HHSR_WAVE) {
            const int ly = p / WS, lx = p - ly * WS;
            const int y = oy + ly, x = ox + lx;
            win[ly * WP + lx] = (y >= 0 && y < mh && x >= 0 && x < mw) ? mov[(size_t)y * mov_pitch + x] : 0.f;
        }
    }
    __syncthreads();
    if (!active) return;
    float rc[PPT], lgx[PPT], lgy[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = lane + k * HHSR_WAVE;
        const size_t o = (size_t)(ty * TS + p / TS) * ref_pitch + tx * TS + (p % TS);
        rc[k] = ref[o];
        lgx[k] = gx[o];
        lgy[k] = gy[o];
    }
    for (int it = 0; it < n_iter; ++it) {
        const float tx_ = truncf(fxv), ty_ = truncf(fyv);
        const float frx = fxv - tx_, fry = fyv - ty_;  // signed fraction of modf (D11)
        const int ix = (int)tx_, iy = (int)ty_;
        const int sx = ix - ix0 + ICA_M, sy = iy - iy0 + ICA_M;  // window offset of this iteration
        const bool in_lds = sx >= 0 && sx <= 2 * ICA_M && sy >= 0 && sy <= 2 * ICA_M;  // wave-uniform
        float B0 = 0.f, B1 = 0.f;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int p = lane + k * HHSR_WAVE;
            const int i = p / TS, j = p % TS;
            float m00, m01, m10, m11;
            if (in_lds) {
                const float* w = win + (i + sy) * WP + (j + sx);
                m00 = w[0];
                m01 = w[1];
                m10 = w[WP];
                m11 = w[WP + 1];
            } else {
                const int x0 = tx * TS + j + ix, y0 = ty * TS + i + iy;
                const bool xa = x0 >= 0 && x0 < mw, xb = x0 + 1 >= 0 && x0 + 1 < mw;
                const bool ya = y0 >= 0 && y0 < mh, yb = y0 + 1 >= 0 && y0 + 1 < mh;
                m00 = (ya && xa) ? mov[(size_t)y0 * mov_pitch + x0] : 0.f;
                m01 = (ya && xb) ? mov[(size_t)y0 * mov_pitch + x0 + 1] : 0.f;
                m10 = (yb && xa) ? mov[(size_t)(y0 + 1) * mov_pitch + x0] : 0.f;
                m11 = (yb && xb) ? mov[(size_t)(y0 + 1) * mov_pitch + x0 + 1] : 0.f;
            }
            ica_tap(m00, m01, m10, m11, frx, fry, rc[k], lgx[k], lgy[k], B0, B1);
        }
        B0 = wave_allsum(B0);
        B1 = wave_allsum(B1);
        fxv = fxv + det_inv * (A11 * B0 - A01 * B1);
        fyv = fyv + det_inv * (-A10 * B0 + A00 * B1);
    }
    if (lane == 0) {
        flow[(size_t)tile * 2] = fxv;
        flow[(size_t)tile * 2 + 1] = fyv;
    }
}

extern "C" int hhsr_ica(const float* ref, const float* gx, const float* gy, int ref_pitch, const float* hess,
                        const float* mov, int mh, int mw, int mov_pitch, float* flow, int ny, int nx, int ts,
                        int n_iter, int flags, void* stream) {
    HHSR_ARG(ref && gx && gy && hess && mov && flow && mh > 0 && mw > 0 && ny > 0 && nx > 0 && n_iter > 0);
    const dim3 grid(nx, ny);
    hipStream_t s = (hipStream_t)stream;
    const int bug = flags & 1;
#define ICA_ARGS ref, gx, gy, ref_pitch, hess, mov, mh, mw, mov_pitch, flow, nx, n_iter, bug
    const int ntiles = nx * ny;
#define ICA_WARGS ref, gx, gy, ref_pitch, hess, mov, mh, mw, mov_pitch, flow, nx, ntiles, n_iter
    switch (ts) {
        case 8: hipLaunchKernelGGL((k_ica<8, 64>), grid, dim3(64), 0, s, ICA_ARGS); break;
        case 16: hipLaunchKernelGGL((k_ica_wave<16>), dim3(hhsr_cdiv(ntiles, 4)), dim3(256), 0, s, ICA_WARGS); break;
        case 32: hipLaunchKernelGGL((k_ica_wave<32>), dim3(hhsr_cdiv(ntiles, 4)), dim3(256), 0, s, ICA_WARGS); break;
        case 64: hipLaunchKernelGGL((k_ica<64, 256>), grid, dim3(256), 0, s, ICA_ARGS); break;
        default:
            hhsr_set_error("hhsr_ica: ICA kernel for tile size %d not implemented", ts);  // ICA.py:100
            return -2;
    }
#undef ICA_ARGS
#undef ICA_WARGS
    HHSR_LAUNCHED();
}

// ---- fused level kernel: block matching + ICA, one wave64 per tile ----------------------------------------
// One launch per pyramid level instead of two (reference alignment.py:125-147).  Each wave stages
//   * its reference tile with a 1-pixel halo (the [-1,0,1] gradients of ICA.py:20-21 are taken from it:
//     zero outside the LEVEL, exactly the values hhsr_grad_hessian writes), and
//   * ONE moving window of (TS + 2r + 2M + 1)^2 pixels around round(flow_in) that serves both the
//     (2r+1)^2 block-matching candidates and the bilinear taps of the ICA iterations,
// i.e. ~3.3 vector loads per pixel for the whole level step (separate kernels: ~7).  The window is staged
// with the block-matching border rule (L2: clamp-to-edge, L1: zero-fill); ICA's own rule (zero outside /
// clamped coordinates for TS = 8) differs only where the window leaves the moving level, and there
// (wave-uniform test) the ICA taps are read from global memory with the exact rule.
// Instruction diet (the level-0 launch is VALU-issue bound: 47k tiles x ~1000 VALU instructions before):
//   * the wave index is made scalar (readfirstlane), so tile coordinates, window origins, flows and every
//     border test live in SGPRs and branch on the scalar unit;
//   * windows are staged through a 2-D lane mapping (lane -> column, 64/CW rows per pass): no per-element
//     divisions, and for tiles whose windows lie inside their level (wave-uniform test) no per-element bounds
//     logic either — one 64-bit add per load, LDS stores at immediate offsets;
//   * the waves of a workgroup never share LDS data: a wave-level fence replaces __syncthreads();
//   * the (2r+1)^2 <= 9 candidate loop is unrolled (immediate LDS offsets).
// a <- [a.lo + a.hi | b.lo + b.hi] over the two 32-lane halves (v_permlane32_swap: vdst[32..63] <-> src[0..31])
__device__ __forceinline__ float swap32_add(float a, float b) {
    const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const unsigned x = t[0], y = t[1];  // (a bit_cast of the vector elements themselves is mis-compiled: x + x)
    return __uint_as_float(x) + __uint_as_float(y);
}
// rows [a.r0 + a.r1, b.r0 + b.r1, a.r2 + a.r3, b.r2 + b.r3] (v_permlane16_swap: odd rows of vdst <-> even rows of src)
__device__ __forceinline__ float swap16_add(float a, float b) {
    const auto t = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const unsigned x = t[0], y = t[1];
    return __uint_as_float(x) + __uint_as_float(y);
}

#ifndef HHSR_ALIGN_PERM
#define HHSR_ALIGN_PERM 1  // level 0: conflict-free window pitch + row permutation (0: round 3's layout, A/B)
#endif
template <int N, int CW>
struct Stage2D {  // N x N window, lane -> (row lane / CW + k * (64 / CW), column lane % CW)
    static constexpr int RPP = HHSR_WAVE / CW, NK = (N + RPP - 1) / RPP;
};

struct AlignFrames {  // blockIdx.y = frame of the batch (the frames share the reference level)
    const float* mov[HHSR_MAX_BATCH];
    float* flow[HHSR_MAX_BATCH];
    const float2* coarse[HHSR_MAX_BATCH];  // all NULL or all set
};

template <int TS, int R, bool L1>
__global__ void __launch_bounds__(256) k_align_wave(const float* __restrict__ ref, int rh, int rw, int ref_pitch,
                                                     const float* __restrict__ hess, AlignFrames fr,
                                                     int mh, int mw, int mov_pitch, int nx,
                                                     int ntiles, int mode, int n_iter,
                                                     int cny, int cnx, int rep,
                                                     float mult) {
    constexpr int r = R;
    const float* __restrict__ mov = fr.mov[blockIdx.y];
    float* __restrict__ flow = fr.flow[blockIdx.y];
    const float2* __restrict__ coarse = fr.coarse[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int M = ICA_M;
    constexpr int RS = TS + 2, RP = RS | 1;            // reference tile + halo
    constexpr int PPT = TS * TS / HHSR_WAVE > 0 ? TS * TS / HHSR_WAVE : 1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / HHSR_WAVE), lane = threadIdx.x & (HHSR_WAVE - 1);
    constexpr int n1 = 2 * r + 1, n = n1 * n1;
    // moving window.  Pitch: odd in general; level 0 (TS = 16, r = 1: 23 x 23) uses 24 together with the row
    // permutation of `li` below — a half-wave then reads 16 + 16 consecutive floats of rows i and i + 2, 2 x 24 = 48 = 16
    // (mod 32) banks apart: conflict-free (pitch 23, rows i and i + 1: 7 of the 32 banks are hit twice by EVERY
    // bilinear tap / candidate read; 46 % of the kernel's LDS cycles were conflicts, profiles/r03_kernel_bottlenecks.md)
    constexpr bool PERM = HHSR_ALIGN_PERM && TS == 16 && R == 1;
    constexpr int WS = TS + 2 * r + 2 * M + 1, WP = PERM ? ((WS + 7) & ~15) + 8 : (WS | 1);
    constexpr int slice = (RS * RP + WS * WP + 3) & ~3;
    float* s_ref = lds + (size_t)wave * slice;
    float* s_win = s_ref + RS * RP;
    const int tile = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;  // neighbouring tiles share window halos: same L2
    if (tile >= ntiles) return;  // wave-uniform; no workgroup barrier below
    const int ty = tile / nx, tx = tile - ty * nx;
    float* fl = flow + (size_t)tile * 2;
    // incoming flow: this level's array, or (fused nearest-neighbour upscaling, alignment.py:150-172) the coarser
    // level's flow of tile (ty/rep, tx/rep) times the level factor — zero past the coarse grid — or zero (rep < 0)
    float fin0 = 0.f, fin1 = 0.f;
    if (coarse) {
        const int sy = ty / rep, sx = tx / rep;
        if (sy < cny && sx < cnx) {
            const float2 c = coarse[(size_t)sy * cnx + sx];
            fin0 = c.x * mult;
            fin1 = c.y * mult;
        }
    } else if (rep >= 0) {
        fin0 = fl[0];
        fin1 = fl[1];
    }
    const float f0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fin0)));
    const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fin1)));
    const float r0 = rintf(f0), r1 = rintf(f1);  // round-half-even
    const int ox = tx * TS + (int)r0 - r - M, oy = ty * TS + (int)r1 - r - M;  // window origin in the moving level
    {
        // all global loads of the two windows are issued back to back into registers, then stored to LDS
        // (a load -> store loop serialises on memory latency: 14-40 round trips per tile)
        constexpr int CWR = RS <= 32 ? 32 : 64, CWW = WS <= 32 ? 32 : 64;
        using SR = Stage2D<RS, CWR>;
        using SW = Stage2D<WS, CWW>;
        float vr[SR::NK], vw[SW::NK];
        const int jr = lane % CWR, ir = lane / CWR, jw = lane % CWW, iw = lane / CWW;
        const int rx0 = tx * TS - 1, ry0 = ty * TS - 1;
        const bool ref_in = rx0 >= 0 && ry0 >= 0 && rx0 + RS <= rw && ry0 + RS <= rh;
        const bool win_in = ox >= 0 && oy >= 0 && ox + WS <= mw && oy + WS <= mh;
        // (window origins are wave-uniform: the row base advances in SGPRs, every load of a window shares ONE 32-bit lane
        // offset — `global_load ... v_off, s[base:base+1]` — instead of a 64-bit address per load; round 6)
        if (ref_in) {  // columns past the window re-read its last column (never stored)
            const char* qb = reinterpret_cast<const char*>(ref + (size_t)ry0 * ref_pitch + rx0);
            const unsigned voff = (unsigned)(ir * ref_pitch + min(jr, RS - 1)) * 4u;
#pragma unroll
            for (int k = 0; k < SR::NK; ++k) {
                const bool tail = (k + 1) * SR::RPP > RS;  // compile time: this pass can run past the last row
                vr[k] = (!tail || ir + k * SR::RPP < RS)
                            ? *reinterpret_cast<const float*>(qb + (size_t)k * SR::RPP * ref_pitch * 4 + voff) : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SR::NK; ++k) {
                const int y = ry0 + ir + k * SR::RPP, x = rx0 + jr;
                vr[k] = (jr < RS && ir + k * SR::RPP < RS && y >= 0 && y < rh && x >= 0 && x < rw)
                            ? ref[(size_t)y * ref_pitch + x] : 0.f;
            }
        }
        if (win_in) {
            const char* qb = reinterpret_cast<const char*>(mov + (size_t)oy * mov_pitch + ox);
            const unsigned voff = (unsigned)(iw * mov_pitch + min(jw, WS - 1)) * 4u;
#pragma unroll
            for (int k = 0; k < SW::NK; ++k) {
                const bool tail = (k + 1) * SW::RPP > WS;
                vw[k] = (!tail || iw + k * SW::RPP < WS)
                            ? *reinterpret_cast<const float*>(qb + (size_t)k * SW::RPP * mov_pitch * 4 + voff) : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SW::NK; ++k) {
                const int y = oy + iw + k * SW::RPP, x = ox + jw;
                const bool inw = jw < WS && iw + k * SW::RPP < WS;
                if (L1) vw[k] = (inw && y >= 0 && y < mh && x >= 0 && x < mw) ? mov[(size_t)y * mov_pitch + x] : 0.f;
                else vw[k] = inw ? mov[(size_t)clampi(y, 0, mh - 1) * mov_pitch + clampi(x, 0, mw - 1)] : 0.f;
            }
        }
        if (jr < RS) {
            float* d = s_ref + ir * RP + jr;
#pragma unroll
            for (int k = 0; k < SR::NK; ++k)
                if ((k + 1) * SR::RPP <= RS || ir + k * SR::RPP < RS) d[k * SR::RPP * RP] = vr[k];
        }
        if (jw < WS) {
            float* d = s_win + iw * WP + jw;
#pragma unroll
            for (int k = 0; k < SW::NK; ++k)
                if ((k + 1) * SW::RPP <= WS || iw + k * SW::RPP < WS) d[k * SW::RPP * WP] = vw[k];
        }
    }
    // LDS operations of one wave complete in order; the slice is private to the wave
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane's pixels p = lane + 64k: row lane / TS + k * (64 / TS), column lane % TS
    constexpr int RSTEP = HHSR_WAVE / TS > 0 ? HHSR_WAVE / TS : 1;
    // (PERM: the four 16-lane groups own rows 0, 2, 1, 3 (+ 4k) of the tile)
    const int li = PERM ? (((lane >> 4) & 1) << 1) | (lane >> 5) : lane / TS, lj = lane % TS;
    // ---------------- block matching ----------------
    float nfx = f0, nfy = f1;  // flow after block matching
    if (L1 && mode == 1) {     // "L1_ref_effective": flow <- round(flow)
        nfx = r0;
        nfy = r1;
    } else {
        CostIdx best{INFINITY, 0};
        if (n <= 16 && TS * TS >= HHSR_WAVE) {
            float rv[PPT];
            const float* rb = s_ref + (li + 1) * RP + lj + 1;
            const float* wb = s_win + (li + M) * WP + lj + M;
#pragma unroll
            for (int k = 0; k < PPT; ++k) rv[k] = rb[k * RSTEP * RP];
#pragma unroll
            for (int c = 0; c < n; ++c) {
                const int dy = c / n1, dx = c - dy * n1;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const float d = rv[k] - wb[(k * RSTEP + dy) * WP + dx];
                    acc += L1 ? fabsf(d) : d * d;
                }
                acc = wave_sum_uniform(acc);
                if (acc < best.c) {
                    best.c = acc;
                    best.i = c;
                }
            }
        } else {
            // (2r+1)^2 = 25 or 81 candidates.  A lane owns PPT vertically adjacent pixels of one column, so a
            // window value serves up to PPT (pixel, dy) pairs from a register: (PPT + 2r)(2r + 1) LDS reads per
            // lane instead of 2 per difference, every candidate's partial cost in its own VGPR.
            constexpr int NQ = (n + 3) / 4;
            float acc[4 * NQ];
#pragma unroll
            for (int c = 0; c < 4 * NQ; ++c) acc[c] = 0.f;
            const int lc = lane % TS, lr = (lane / TS) * PPT;
            float rv[PPT];
            const float* rb = s_ref + (lr + 1) * RP + lc + 1;
            const float* wb = s_win + (lr + M) * WP + lc + M;
#pragma unroll
            for (int k = 0; k < PPT; ++k) rv[k] = rb[k * RP];
#pragma unroll
            for (int rr = 0; rr < PPT + 2 * r; ++rr) {
#pragma unroll
                for (int dx = 0; dx < n1; ++dx) {
                    const float w = wb[rr * WP + dx];
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int dy = rr - k;
                        if (dy >= 0 && dy < n1) {
                            const float d = rv[k] - w;
                            acc[dy * n1 + dx] = L1 ? acc[dy * n1 + dx] + fabsf(d) : fmaf(d, d, acc[dy * n1 + dx]);
                        }
                    }
                }
            }
            // 4 candidates per step: the lane-half and row swaps of gfx950 fold the 64 partials of candidates
            // (4q, 4q+1, 4q+2, 4q+3) into the 16-lane rows (0, 2, 1, 3) of ONE register, a 4-step DPP row
            // reduction finishes all four at once (10 instructions per 4 candidates, fixed association).
            const int roff = ((lane >> 4) & 1) * 2 + (lane >> 5);  // rows 0..3 hold candidates 4q + {0, 2, 1, 3}
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float s01 = swap32_add(acc[4 * q], acc[4 * q + 1]);      // [c0 | c1]
                const float s23 = swap32_add(acc[4 * q + 2], acc[4 * q + 3]);  // [c2 | c3]
                float v = swap16_add(s01, s23);                                // rows [c0, c2, c1, c3]
                v += HHSR_DPP(v, 0xB1, 0xf);
                v += HHSR_DPP(v, 0x4E, 0xf);
                v += HHSR_DPP(v, 0x141, 0xf);
                v += HHSR_DPP(v, 0x140, 0xf);
                const int idx = 4 * q + roff;
                const float cost = idx < n ? v : INFINITY;
                if (cost < best.c) {  // idx grows with q: strict < keeps the first minimum of this row
                    best.c = cost;
                    best.i = idx;
                }
            }
            best = wave_argmin(best);
        }
        const int dy = best.i / n1 - r, dx = best.i % n1 - r;
        if (L1) {  // flow <- round(flow) + shift
            nfx = r0 + (float)dx;
            nfy = r1 + (float)dy;
        } else {   // shift added to the UN-rounded flow
            nfx = f0 + (float)dx;
            nfy = f1 + (float)dy;
        }
    }
    // ---------------- ICA ----------------
    const float* h = hess + (size_t)tile * 4;
    const float A00 = h[0], A01 = h[1], A10 = h[2], A11 = h[3];
    const float det = A00 * A11 - A01 * A10;
    float fxv = nfx, fyv = nfy;
    if (!(fabsf(det) < 1e-10f)) {  // else: not solvable, the block-matching result stands (ICA.py:124-125)
        const float det_inv = 1.0f / det;
        float rc[PPT], lgx[PPT], lgy[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const float* c = s_ref + (li + k * RSTEP + 1) * RP + lj + 1;  // pixel (li + k RSTEP, lj): the lane's k-th pixel
            rc[k] = c[0];
            lgx[k] = c[1] - c[-1];
            lgy[k] = c[RP] - c[-RP];
        }
        for (int it = 0; it < n_iter; ++it) {
            const float tx_ = truncf(fxv), ty_ = truncf(fyv);
            const float frx = fxv - tx_, fry = fyv - ty_;  // signed fraction of modf (D11)
            const int ix = (int)tx_, iy = (int)ty_;
            // LDS coordinates of the tile's first tap; usable when all taps are inside the staged window AND
            // the tapped region lies inside the moving level (where every border rule agrees)
            const int sx = tx * TS + ix - ox, sy = ty * TS + iy - oy;
            const bool in_lds = sx >= 0 && sy >= 0 && sx + TS + 1 <= WS && sy + TS + 1 <= WS &&
                                tx * TS + ix >= 0 && ty * TS + iy >= 0 && tx * TS + ix + TS < mw &&
                                ty * TS + iy + TS < mh;
            float B0 = 0.f, B1 = 0.f;
            const float* wl = s_win + (li + sy) * WP + (lj + sx);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int i = li + k * RSTEP, j = lj;
                float m00, m01, m10, m11;
                if (in_lds) {
                    const float* w = TS * TS >= HHSR_WAVE ? wl + k * RSTEP * WP : s_win + (i + sy) * WP + (j + sx);
                    m00 = w[0];
                    m01 = w[1];
                    m10 = w[WP];
                    m11 = w[WP + 1];
                } else if (TS == 8) {  // clamped coordinates (ICA.py:152-156)
                    const int x0 = clampi(tx * TS + j + ix, 0, mw - 1), y0 = clampi(ty * TS + i + iy, 0, mh - 1);
                    const int x1 = clampi(x0 + 1, 0, mw - 1), y1 = clampi(y0 + 1, 0, mh - 1);
                    m00 = mov[(size_t)y0 * mov_pitch + x0];
                    m01 = mov[(size_t)y0 * mov_pitch + x1];
                    m10 = mov[(size_t)y1 * mov_pitch + x0];
                    m11 = mov[(size_t)y1 * mov_pitch + x1];
                } else {  // zero outside (ICA.py:240-243)
                    const int x0 = tx * TS + j + ix, y0 = ty * TS + i + iy;
                    const bool xa = x0 >= 0 && x0 < mw, xb = x0 + 1 >= 0 && x0 + 1 < mw;
                    const bool ya = y0 >= 0 && y0 < mh, yb = y0 + 1 >= 0 && y0 + 1 < mh;
                    m00 = (ya && xa) ? mov[(size_t)y0 * mov_pitch + x0] : 0.f;
                    m01 = (ya && xb) ? mov[(size_t)y0 * mov_pitch + x0 + 1] : 0.f;
                    m10 = (yb && xa) ? mov[(size_t)(y0 + 1) * mov_pitch + x0] : 0.f;
                    m11 = (yb && xb) ? mov[(size_t)(y0 + 1) * mov_pitch + x0 + 1] : 0.f;
                }
                ica_tap(m00, m01, m10, m11, frx, fry, rc[k], lgx[k], lgy[k], B0, B1);
            }
            B0 = wave_allsum(B0);
            B1 = wave_allsum(B1);
            fxv = fxv + det_inv * (A11 * B0 - A01 * B1);
            fyv = fyv + det_inv * (-A10 * B0 + A00 * B1);
        }
    }
    if (lane == 0) {
        fl[0] = fxv;
        fl[1] = fyv;
    }
}

static size_t align_wave_lds(int ts, int r) {
    const int RS = ts + 2, RP = RS | 1, WS = ts + 2 * r + 2 * ICA_M + 1;
    const int WP = (HHSR_ALIGN_PERM && ts == 16 && r == 1) ? ((WS + 7) & ~15) + 8 : (WS | 1);  // (k_align_wave's PERM pitch)
    return (size_t)4 * ((RS * RP + WS * WP + 3) & ~3) * sizeof(float);
}

extern "C" int hhsr_align_level_batch(const float* ref, int rh, int rw, int ref_pitch, const float* hess,
                                      const float* const* movs, int n_frames, int mh, int mw, int mov_pitch,
                                      float* const* flows, int ny, int nx, int ts, int r, int metric, int n_iter,
                                      const float* const* coarse_flows, int cny, int cnx, int rep, float mult,
                                      void* stream) {
    HHSR_ARG(ref && hess && movs && flows && n_frames >= 0 && rh > 0 && rw > 0 && mh > 0 && mw > 0 && ny > 0 && nx > 0);
    for (int n = 0; n < n_frames; ++n) HHSR_ARG(movs[n] && flows[n] && (!coarse_flows || coarse_flows[n]));
    HHSR_ARG(r >= 0 && n_iter > 0 && metric >= 0 && metric <= 2);  // 0 = L2, 1 = L1 (intended), 2 = L1_ref_effective
    HHSR_ARG(ts == 8 || ts == 16 || ts == 32);
    HHSR_ARG(metric == 0 || ts >= 16);  // block_matching.py:87: no L1 search for 8-pixel tiles
    HHSR_ARG(ny * ts <= rh && nx * ts <= rw);
    HHSR_ARG(r == 1 || r == 2 || r == 4);  // compiled search radii (other radii: hhsr_bm_* + hhsr_ica)
    HHSR_ARG(!coarse_flows || (cny > 0 && cnx > 0 && rep > 0));
    const size_t l = align_wave_lds(ts, r);
    HHSR_ARG(l <= 64 * 1024);
    const int ntiles = nx * ny;
    const dim3 b(256);
    hipStream_t s = (hipStream_t)stream;
    const int mode = metric == 2 ? 1 : 0;
    for (int n0 = 0; n0 < n_frames; n0 += HHSR_MAX_BATCH) {
        const int nb = n_frames - n0 < HHSR_MAX_BATCH ? n_frames - n0 : HHSR_MAX_BATCH;
        AlignFrames fr;
        for (int k = 0; k < HHSR_MAX_BATCH; ++k) {
            const int n = n0 + (k < nb ? k : 0);
            fr.mov[k] = movs[n];
            fr.flow[k] = flows[n];
            fr.coarse[k] = coarse_flows ? reinterpret_cast<const float2*>(coarse_flows[n]) : nullptr;
        }
        const dim3 g(hhsr_cdiv(ntiles, 4), nb);
#define ALW(TS, R, L1) hipLaunchKernelGGL((k_align_wave<TS, R, L1>), g, b, l, s, ref, rh, rw, ref_pitch, hess, fr, mh, \
                                          mw, mov_pitch, nx, ntiles, mode, n_iter, cny, cnx, rep, mult)
#define ALW_R(TS, L1) do { if (r == 1) ALW(TS, 1, L1); else if (r == 2) ALW(TS, 2, L1); else ALW(TS, 4, L1); } while (0)
        if (metric == 0) {
            if (ts == 8) ALW_R(8, false);
            else if (ts == 16) ALW_R(16, false);
            else ALW_R(32, false);
        } else {
            if (ts == 16) ALW_R(16, true);
            else ALW_R(32, true);
        }
#undef ALW_R
#undef ALW
    }
    HHSR_LAUNCHED();
}

extern "C" int hhsr_align_level(const float* ref, int rh, int rw, int ref_pitch, const float* hess, const float* mov,
                                int mh, int mw, int mov_pitch, float* flow, int ny, int nx, int ts, int r, int metric,
                                int n_iter, const float* coarse_flow, int cny, int cnx, int rep, float mult,
                                void* stream) {
    HHSR_ARG(mov && flow);
    return hhsr_align_level_batch(ref, rh, rw, ref_pitch, hess, &mov, 1, mh, mw, mov_pitch, &flow, ny, nx, ts, r, metric,
                                  n_iter, coarse_flow ? &coarse_flow : nullptr, cny, cnx, rep, mult, stream);
}

// ---- flow upscaling (nearest) ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_flow_upscale(const float2* __restrict__ src, int sny, int snx,
                                                       float2* __restrict__ dst, int dny, int dnx, int rep,
                                                       float mult) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dnx) return;
    const int sy = y / rep, sx = x / rep;
    float2 v = make_float2(0.f, 0.f);
    if (sy < sny && sx < snx) {
        v = src[(size_t)sy * snx + sx];
        v.x *= mult;
        v.y *= mult;
    }
    dst[(size_t)y * dnx + x] = v;
}

extern "C" int hhsr_flow_upscale_nearest(const float* src, int sny, int snx, float* dst, int dny, int dnx, int rep,
                                         float mult, void* stream) {
    HHSR_ARG(src && dst && sny > 0 && snx > 0 && dny > 0 && dnx > 0 && rep >= 1);
    hipLaunchKernelGGL(k_flow_upscale, dim3(hhsr_cdiv(dnx, 256), dny), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(src), sny, snx, reinterpret_cast<float2*>(dst), dny, dnx, rep,
                       mult);
    HHSR_LAUNCHED();
}

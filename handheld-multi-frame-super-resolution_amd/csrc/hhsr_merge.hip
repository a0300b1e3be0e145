// Alg. 4 / Alg. 11 anisotropic kernel-regression merge (reference merge.py:22-434, linalg.py:38-200).
//
// One thread per high-resolution output pixel, 64x4 pixel workgroups (a wave64 covers 64 consecutive
// pixels of one output row, so the [sH][sW][3] accumulators are read/written as contiguous 768-byte
// runs).  Two entry shapes:
//   hhsr_accumulate / hhsr_accumulate_ref   per-frame read-modify-write of num/den, the reference's
//                                           operator API (2 x 12 S P bytes of accumulator traffic per frame)
//   hhsr_merge_burst                        loops over all resident frames with the accumulators in
//                                           registers and writes the output once: the accumulator
//                                           traffic drops from 48 S P bytes per frame to 12-24 S P per burst.
// Arithmetic follows the reference's Numba typing: coordinates, covariance
// interpolation and weights are float64, the per-pixel val/acc sums are float32 rounded after every
// tap.  `WT` selects the type of the weight chain: double = the reference's typing (validation mode,
// HHSR_WEIGHT_F64), float = the default fast path (comp_accum_fast).
#include "hhsr_merge.h"

// ---- per-frame kernels (operator API) ----------------------------------------------------------------
template <typename WT, int GEOM, bool ISO>
__global__ void __launch_bounds__(256) k_accumulate(FramePtr f, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                     float* __restrict__ den) {
    const int hj = blockIdx.x * 64 + (threadIdx.x & 63), hi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (hj >= g.sW || hi >= g.sH) return;
    float val[3] = {0.f, 0.f, 0.f}, acc[3] = {0.f, 0.f, 0.f};
    if (sizeof(WT) == 4 && !border_pixel(g, hi, hj)) {
        const Pix p = make_pix(g, hi, hj);
        float n4[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, d4[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        comp_accum_fast<GEOM, ISO>(f, g, p, n4, d4);
        classes_to_rgb(cfa, n4, d4, val, acc);
    } else {  // float64 weight chain: validation mode, and always on the border bands (see border_pixel)
        comp_contrib<double, ISO>(f, g, cfa, hi, hj, val, acc);
    }
    const size_t o = ((size_t)hi * g.sW + hj) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        num[o + k] += val[k];
        den[o + k] += acc[k];
    }
}

template <bool ISO>
__global__ void __launch_bounds__(256) k_accumulate_ref(const float* __restrict__ raw,
                                                         const float4* __restrict__ cov, Geo g, Cfa4 cfa,
                                                         const float* __restrict__ acc_rob, int rad_max,
                                                         double max_mult, double max_fc, float* __restrict__ num,
                                                         float* __restrict__ den, int divide, int fast) {
    const int oj = blockIdx.x * 64 + (threadIdx.x & 63), oi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (oj >= g.sW || oi >= g.sH) return;
    float val[3] = {0.f, 0.f, 0.f}, acc[3] = {0.f, 0.f, 0.f};
    bool over = false, plain = fast && !border_pixel(g, oi, oj);
    if (plain && acc_rob) {  // (ref_contrib's test: the denoiser widens / overwrites where few frames were merged)
        const float pyf = (float)((double)(oi + g.off_hr) / g.scale) - (float)g.off_lr, pxf = (float)((double)oj / g.scale);
        const int ry_i = min((int)rintf(pyf), g.H - 1), rx_i = min((int)rintf(pxf), g.W - 1);
        plain = !((double)acc_rob[(size_t)ry_i * g.W + rx_i] <= max_fc);
    }
    if (plain) {
        // HHSR_REF_FAST: float32 weights for the pixels the denoiser leaves alone, outside the border bands — the fused
        // merge's reference frame (merge_pixel / k_merge_x2); the float64 chain below is 1.5 ms per 12 MP x2 pass
        float n4[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, d4[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        ref_accum_fast<ISO>(raw, cov, g, oi, oj, n4, d4);
        classes_to_rgb(cfa, n4, d4, val, acc);
    } else {
        over = ref_contrib<ISO>(raw, cov, g, cfa, oi, oj, acc_rob, rad_max, max_mult, max_fc, val, acc);
    }
    const size_t o = ((size_t)oi * g.sW + oj) * 3;
    if (divide) {  // HHSR_REF_DIVIDE: the normalisation (utils.py:62-90) in the same pass; den stays as it was
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float nk = num[o + k], dk = den[o + k];
            if (!(g.mono && k > 0)) {  // (one channel: the others are divided as they are)
                nk = over ? val[k] : nk + val[k];
                dk = over ? acc[k] : dk + acc[k];
            }
            num[o + k] = nk / dk;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (g.mono && k > 0) break;  // one channel: the others are not even overwritten (merge.py:223-233)
        if (over) {
            num[o + k] = val[k];
            den[o + k] = acc[k];
        } else {
            num[o + k] += val[k];
            den[o + k] += acc[k];
        }
    }
}


template <typename WT, int GEOM, bool ISO>
__global__ void __launch_bounds__(256) k_merge_burst(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                      float* __restrict__ den) {
    const int hj = blockIdx.x * 64 + (threadIdx.x & 63), hi = g.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (hj >= g.sW || hi >= g.row1) return;
    if (owns_lr_pixel(a, hi, hj)) {
        const Pix p = make_pix(g, hi, hj);
        float racc = (a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[p.ridx] : 0.f;
        for (int n = 0; n < a.n; ++n) racc += a.f[n].r[p.ridx];
        a.acc_r[p.ridx] = racc;
    }
    if (sizeof(WT) == 4 && border_pixel(g, hi, hj)) return;  // k_merge_border's
    merge_pixel<WT, GEOM, ISO>(a, g, cfa, hi, hj, num, den);
}

// The border bands (see border_pixel) with the reference's float64 weight chain.  Threads enumerate the bt + bb full
// rows first, then the bl + br columns of the rows in between (no pixel twice: LOAD_ACC reads what it overwrites).
template <bool ISO>
__global__ void __launch_bounds__(256) k_merge_border(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                       float* __restrict__ den) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int nrow = g.bt + g.bb, ncol = g.bl + g.br, mid = g.sH - nrow;
    int hi, hj;
    if (t < nrow * g.sW) {
        const int r = t / g.sW;
        hj = t - r * g.sW;
        hi = r < g.bt ? r : g.sH - g.bb + (r - g.bt);
    } else {
        const int u = t - nrow * g.sW;
        if (ncol == 0 || u >= mid * ncol) return;
        const int r = u / ncol, c = u - r * ncol;
        hi = g.bt + r;
        hj = c < g.bl ? c : g.sW - g.br + (c - g.bl);
    }
    if (hi < g.row0 || hi >= g.row1) return;
    merge_pixel<double, GEOM_F64, ISO>(a, g, cfa, hi, hj, num, den);
}

// The same with the frames of a pixel spread over lanes.  One thread per border pixel walks its 20 frames through the
// float64 chain one after the other: ~70 k threads (1 100 waves on 1 024 SIMDs) with 15 k instructions each — the
// launch is pure latency (158 us at 12 MP x 20).  Here lane = (pixel, frame): every lane evaluates ONE frame's
// contribution to its pixel (the reference frame is the last "frame"), then the pixel's first lane adds the
// contributions in frame order — the float32 additions of merge_pixel, in the same order: bit-identical.
template <bool ISO>
__global__ void __launch_bounds__(256) k_merge_border_wave(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                            float* __restrict__ den, int nf, int ppw) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pl = lane / nf, f = lane - pl * nf;  // pixel slot of the wave, frame (f == a.n: the reference frame)
    const int t = wave * ppw + pl;
    const int nrow = g.bt + g.bb, ncol = g.bl + g.br, mid = g.sH - nrow;
    int hi = -1, hj = 0;
    if (pl < ppw) {
        if (t < nrow * g.sW) {
            const int r = t / g.sW;
            hj = t - r * g.sW;
            hi = r < g.bt ? r : g.sH - g.bb + (r - g.bt);
        } else if (ncol > 0 && t - nrow * g.sW < mid * ncol) {
            const int u = t - nrow * g.sW;
            const int r = u / ncol, c = u - r * ncol;
            hi = g.bt + r;
            hj = c < g.bl ? c : g.sW - g.br + (c - g.bl);
        }
    }
    const bool live = hi >= g.row0 && hi < g.row1;
    float val[3] = {0.f, 0.f, 0.f}, acc[3] = {0.f, 0.f, 0.f};
    if (live) {
        if (f < a.n) comp_contrib<double, ISO>(a.f[f], g, cfa, hi, hj, val, acc, (a.flags & HHSR_MERGE_LOCAL_MIN) != 0);
        else ref_contrib<ISO>(a.ref_raw, a.ref_cov, g, cfa, hi, hj, nullptr, 0, 0.0, 0.0, val, acc);
    }
    const size_t o = live ? ((size_t)(hi - g.row0) * g.sW + hj) * 3 : 0;
    float n3[3] = {0.f, 0.f, 0.f}, d3[3] = {0.f, 0.f, 0.f};
    if (live && f == 0 && (a.flags & HHSR_MERGE_LOAD_ACC)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n3[k] = num[o + k];
            d3[k] = den[o + k];
        }
    }
    for (int q = 0; q < nf; ++q) {  // wave-uniform trip count; lane (pl, 0) accumulates in frame order
        const int src = pl * nf + q;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n3[k] += __shfl(val[k], src);
            d3[k] += __shfl(acc[k], src);
        }
    }
    if (live && f == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            num[o + k] = (a.flags & HHSR_MERGE_DIVIDE) ? n3[k] / d3[k] : n3[k];
            if (a.flags & HHSR_MERGE_STORE_DEN) den[o + k] = d3[k];
        }
    }
}



static bool scale_is_pow2(double s) {  // 1, 2, 4, 8: (h + 0.5)/s is exact in float32
    return s == 1.0 || s == 2.0 || s == 4.0 || s == 8.0;
}

static int fill_geo(Geo& g, int H, int W, int pitch, int ny, int nx, int ts, double scale, int sH, int sW, bool mono) {
    g.H = H; g.W = W; g.pitch = pitch; g.mono = mono ? 1 : 0;
    g.gh = mono ? H : H / 2; g.gw = mono ? W : W / 2;
    g.ny = ny; g.nx = nx; g.ts = ts; g.sH = sH; g.sW = sW; g.scale = scale;
    g.row0 = 0; g.row1 = sH;
    g.off_lr = g.off_hr = 0;
    // border bands: output rows / columns whose reference-window centre rint(float(idx / scale)) (merge.py:113-114,
    // 179-180) is the first or last raw row / column
    auto bands = [scale](int n_lr, int n_hr, int& lo, int& hi) {
        lo = 0;
        while (lo < n_hr && (int)rintf((float)((double)lo / scale)) <= 0) ++lo;
        hi = 0;
        while (hi < n_hr - lo && (int)rintf((float)((double)(n_hr - 1 - hi) / scale)) >= n_lr - 1) ++hi;
    };
    bands(H, sH, g.bt, g.bb);
    bands(W, sW, g.bl, g.br);
    return 0;
}

// the border bands of a float32 launch, with the float64 chain (after the main kernel, same stream)
template <class Launch>
static void launch_border(const Geo& g, Launch launch) {
    const int64_t n = (int64_t)(g.bt + g.bb) * g.sW + (int64_t)(g.bl + g.br) * (g.sH - g.bt - g.bb);
    if (n > 0) launch(dim3((unsigned)((n + 255) / 256)), dim3(256));
}


extern "C" int hhsr_accumulate(const float* raw, int H, int W, int pitch, const float* flow, int ny, int nx, int ts,
                               const float* covs, const float* r, const uint8_t cfa[4], double scale, int kflags,
                               float* num, float* den, int sH, int sW, void* stream) {
    const int iso = kflags & HHSR_KERNEL_ISO, f64 = kflags & HHSR_WEIGHT_F64;
    const bool mono = (kflags & HHSR_SENSOR_MONO) != 0;
    HHSR_ARG(raw && flow && r && (cfa || mono) && num && den && (iso || covs));
    HHSR_ARG(H >= 2 && W >= 2 && pitch >= W && ts > 0 && scale >= 1.0 && sH > 0 && sW > 0);
    HHSR_ARG((int64_t)ny * ts >= H && (int64_t)nx * ts >= W);  // every LR position has a flow tile
    HHSR_ARG((double)sH <= scale * H + 0.5 && (double)sW <= scale * W + 0.5);
    for (int k = 0; k < 4 && !mono; ++k) HHSR_ARG(cfa[k] <= 2);
    Geo g;
    fill_geo(g, H, W, pitch, ny, nx, ts, scale, sH, sW, mono);
    Cfa4 c;
    for (int k = 0; k < 4; ++k) c.c[k] = mono ? 0 : cfa[k];
    FramePtr f{raw, reinterpret_cast<const float2*>(flow), reinterpret_cast<const float4*>(covs), r};
    const dim3 grid(hhsr_cdiv(sW, 64), hhsr_cdiv(sH, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool p2 = scale_is_pow2(scale);
#define HHSR_ACC(WT, GEOM, ISO) hipLaunchKernelGGL((k_accumulate<WT, GEOM, ISO>), grid, block, 0, s, f, g, c, num, den)
    if (f64) { if (iso) HHSR_ACC(double, GEOM_F64, true); else HHSR_ACC(double, GEOM_F64, false); }
    else if (p2) { if (iso) HHSR_ACC(float, GEOM_P2, true); else HHSR_ACC(float, GEOM_P2, false); }
    else { if (iso) HHSR_ACC(float, GEOM_F64, true); else HHSR_ACC(float, GEOM_F64, false); }
#undef HHSR_ACC
    HHSR_LAUNCHED();
}

extern "C" int hhsr_accumulate_ref(const float* raw, int H, int W, int pitch, const float* covs,
                                   const uint8_t cfa[4], double scale, int kflags, const float* acc_rob, int rad_max,
                                   double max_multiplier, double max_frame_count, float* num, float* den, int sH,
                                   int sW, void* stream) {
    const int iso = kflags & HHSR_KERNEL_ISO;
    const bool mono = (kflags & HHSR_SENSOR_MONO) != 0;
    HHSR_ARG(raw && (cfa || mono) && num && den && (iso || covs));
    HHSR_ARG(H >= 2 && W >= 2 && pitch >= W && scale >= 1.0 && sH > 0 && sW > 0);
    HHSR_ARG(!acc_rob || (rad_max >= 0 && rad_max <= 8 && max_multiplier > 0.0));
    for (int k = 0; k < 4 && !mono; ++k) HHSR_ARG(cfa[k] <= 2);
    Geo g;
    fill_geo(g, H, W, pitch, 0, 0, 1, scale, sH, sW, mono);
    Cfa4 c;
    for (int k = 0; k < 4; ++k) c.c[k] = mono ? 0 : cfa[k];
    const dim3 grid(hhsr_cdiv(sW, 64), hhsr_cdiv(sH, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const float4* cv = reinterpret_cast<const float4*>(covs);
    if (iso)
        hipLaunchKernelGGL((k_accumulate_ref<true>), grid, block, 0, s, raw, cv, g, c, acc_rob, rad_max,
                           max_multiplier, max_frame_count, num, den, kflags & HHSR_REF_DIVIDE, kflags & HHSR_REF_FAST);
    else
        hipLaunchKernelGGL((k_accumulate_ref<false>), grid, block, 0, s, raw, cv, g, c, acc_rob, rad_max,
                           max_multiplier, max_frame_count, num, den, kflags & HHSR_REF_DIVIDE, kflags & HHSR_REF_FAST);
    HHSR_LAUNCHED();
}

static int merge_burst_impl(const float* const* raws, const float* const* flows, const float* const* covs,
                            const float* const* rs, int n_frames, int H, int W, int pitch, int ny, int nx,
                            int ts, const float* ref_raw, const float* ref_covs, const uint8_t cfa[4],
                            double scale, int kflags, int flags, float* num, float* den, float* acc_r, int sH,
                            int sW, int row0, int nrows, int lr_row_offset, float* class_acc, int n_done, void* stream);

extern "C" int hhsr_merge_burst(const float* const* raws, const float* const* flows, const float* const* covs,
                                const float* const* rs, int n_frames, int H, int W, int pitch, int ny, int nx,
                                int ts, const float* ref_raw, const float* ref_covs, const uint8_t cfa[4],
                                double scale, int kflags, int flags, float* num, float* den, float* acc_r, int sH,
                                int sW, int row0, int nrows, int lr_row_offset, void* stream) {
    if (flags & (HHSR_MERGE_STORE_CLASSES | HHSR_MERGE_LOAD_CLASSES)) {
        hhsr_set_error("hhsr_merge_burst: chained launches go through hhsr_merge_burst_chain");
        return -1;
    }
    return merge_burst_impl(raws, flows, covs, rs, n_frames, H, W, pitch, ny, nx, ts, ref_raw, ref_covs, cfa, scale, kflags,
                            flags, num, den, acc_r, sH, sW, row0, nrows, lr_row_offset, nullptr, 0, stream);
}

extern "C" size_t hhsr_merge_chain_bytes(int H, int W) {
    return (size_t)hhsr_cdiv(W, QT) * (size_t)hhsr_cdiv(H, QT) * 33 * 256 * sizeof(float);
}

extern "C" int hhsr_merge_burst_chain(const float* const* raws, const float* const* flows, const float* const* covs,
                                      const float* const* rs, int n_frames, int H, int W, int pitch, int ny, int nx,
                                      int ts, const float* ref_raw, const float* ref_covs, const uint8_t cfa[4],
                                      double scale, int kflags, int flags, float* num, float* den, float* acc_r, int sH,
                                      int sW, float* class_acc, int n_done, void* stream) {
    const int st = flags & HHSR_MERGE_STORE_CLASSES, ld = flags & HHSR_MERGE_LOAD_CLASSES;
    HHSR_ARG(class_acc && ((uintptr_t)class_acc & 3) == 0 && (st || ld));  // (both: a middle link)
    HHSR_ARG(!st || !(flags & (HHSR_MERGE_DO_REF | HHSR_MERGE_DIVIDE | HHSR_MERGE_LOAD_ACC | HHSR_MERGE_STORE_DEN)));
    HHSR_ARG(!ld || (n_done > 0 && n_done <= n_frames && !(flags & HHSR_MERGE_LOAD_ACC)));
    HHSR_ARG(!(kflags & (HHSR_SENSOR_MONO | HHSR_WEIGHT_F64 | HHSR_MERGE_FORCE_GENERIC | HHSR_MERGE_FORCE_TILE | HHSR_MERGE_FORCE_X2V1)));
    return merge_burst_impl(raws, flows, covs, rs, n_frames, H, W, pitch, ny, nx, ts, ref_raw, ref_covs, cfa, scale, kflags,
                            flags, num, den, acc_r, sH, sW, 0, sH, 0, class_acc, ld ? n_done : 0, stream);
}

static int merge_burst_impl(const float* const* raws, const float* const* flows, const float* const* covs,
                            const float* const* rs, int n_frames, int H, int W, int pitch, int ny, int nx,
                            int ts, const float* ref_raw, const float* ref_covs, const uint8_t cfa[4],
                            double scale, int kflags, int flags, float* num, float* den, float* acc_r, int sH,
                            int sW, int row0, int nrows, int lr_row_offset, float* class_acc, int n_done, void* stream) {
    const int iso = kflags & HHSR_KERNEL_ISO, f64 = kflags & HHSR_WEIGHT_F64;
    const bool mono = (kflags & HHSR_SENSOR_MONO) != 0;
    HHSR_ARG(n_frames >= 0 && n_frames <= HHSR_MAX_FRAMES && (cfa || mono) && num);
    HHSR_ARG(n_frames == 0 || (raws && flows && rs && (iso || covs)));
    HHSR_ARG(H >= 2 && W >= 2 && pitch >= W && ts > 0 && scale >= 1.0 && sH > 0 && sW > 0);
    HHSR_ARG(n_frames == 0 || ((int64_t)ny * ts >= H && (int64_t)nx * ts >= W));
    HHSR_ARG((double)sH <= scale * H + 0.5 && (double)sW <= scale * W + 0.5);
    HHSR_ARG(!(flags & HHSR_MERGE_DO_REF) || (ref_raw && (iso || ref_covs)));
    HHSR_ARG(!(flags & (HHSR_MERGE_LOAD_ACC | HHSR_MERGE_STORE_DEN)) || den);
    for (int k = 0; k < 4 && !mono; ++k) HHSR_ARG(cfa[k] <= 2);
    BurstArgs a;
    for (int n = 0; n < n_frames; ++n) {
        HHSR_ARG(raws[n] && flows[n] && rs[n] && (iso || covs[n]));
        a.f[n] = FramePtr{raws[n], reinterpret_cast<const float2*>(flows[n]),
                          iso ? nullptr : reinterpret_cast<const float4*>(covs[n]), rs[n]};
    }
    for (int n = n_frames; n < HHSR_MAX_FRAMES; ++n) a.f[n] = FramePtr{nullptr, nullptr, nullptr, nullptr};
    HHSR_ARG(!acc_r || ((double)(int)scale == scale && n_frames > 0));  // LR <-> HR pixel ownership needs an integer scale
    a.acc_r = acc_r;
    a.iscale = (int)scale;
    a.n = n_frames;
    a.ref_raw = ref_raw;
    a.ref_cov = reinterpret_cast<const float4*>(ref_covs);
    a.flags = flags;
    a.cls = class_acc;
    a.first = n_done;
    Geo g;
    fill_geo(g, H, W, pitch, ny, nx, ts, scale, sH, sW, mono);
    HHSR_ARG(row0 >= 0 && nrows > 0 && row0 + nrows <= sH);
    HHSR_ARG(lr_row_offset >= 0 && lr_row_offset % ts == 0 && lr_row_offset % 2 == 0 &&
             (double)(int64_t)(lr_row_offset * scale) == lr_row_offset * scale);  // whole tiles, Bayer quads, output rows
    g.off_lr = lr_row_offset;
    g.off_hr = (int)(lr_row_offset * scale);
    g.row0 = row0;
    g.row1 = row0 + nrows;
    Cfa4 c;
    for (int k = 0; k < 4; ++k) c.c[k] = mono ? 0 : cfa[k];
    const dim3 grid(hhsr_cdiv(sW, 64), hhsr_cdiv(nrows, 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool p2 = scale_is_pow2(scale);
    // kernel choice.  LDS-staged tile kernel: integer scale, 16-px HR workgroups inside one flow tile, windows fit the
    // LDS arrays; x2 kernels: one thread per LR pixel (4 HR pixels), 32 x 32 HR workgroups inside one flow tile.
    // kflags HHSR_MERGE_FORCE_* (validation / A-B measurements) restrict the choice; the environment variables
    // HHSR_MERGE_NO_LDS / _NO_QUAD / _X2_V1 do the same for a whole process and are read once.
    static const int env_force = (getenv("HHSR_MERGE_NO_LDS") ? HHSR_MERGE_FORCE_GENERIC : 0) |
                                 (getenv("HHSR_MERGE_NO_QUAD") ? HHSR_MERGE_FORCE_TILE : 0) |
                                 (getenv("HHSR_MERGE_X2_V1") ? HHSR_MERGE_FORCE_X2V1 : 0);
    // monochrome sensors: the x2 tile kernel with a per-pixel covariance window (k_merge_burst_quad<.., MONO>); every other
    // scale takes the generic kernel (the tile / wave-per-class kernels are laid out for the Bayer covariance grid)
    const int force = kflags | env_force;
    const int iscale = (int)scale;
    const bool tiled = !f64 && (double)iscale == scale && iscale >= 1 && ((int64_t)ts * iscale) % MT == 0 &&
                       n_frames > 0 && row0 % MT == 0 && !(force & HHSR_MERGE_FORCE_GENERIC) && !(mono && iscale != 2);
    const bool lmin = (flags & HHSR_MERGE_LOCAL_MIN) != 0;
    const bool quad = tiled && p2 && iscale == 2 && ts % QT == 0 && sW == 2 * W && sH == 2 * H && row0 % (2 * QT) == 0 &&
                      nrows % 2 == 0 && !(force & HHSR_MERGE_FORCE_TILE);
    const bool x2_v1 = (force & HHSR_MERGE_FORCE_X2V1) != 0;
    const bool aligned16 = ((uintptr_t)num % 16 == 0) && (!(flags & HHSR_MERGE_STORE_DEN) || (uintptr_t)den % 16 == 0);
    const bool x3 = aligned16 && tiled && iscale == 3 && cfa_is_bayer(c) && ts % QT == 0 && sW == 3 * W && sH == 3 * H && row0 % (3 * QT) == 0 && nrows % 3 == 0 &&
                    W % 4 == 0 && !(force & HHSR_MERGE_FORCE_TILE);
    const bool chained = (flags & (HHSR_MERGE_STORE_CLASSES | HHSR_MERGE_LOAD_CLASSES)) != 0;
    if (chained && !(quad && !x2_v1 && aligned16 && !mono && cfa_is_bayer(c))) {
        hhsr_set_error("hhsr_merge_burst_chain: needs the wave-per-class x2 kernel (scale 2, ts %% 16 == 0, sH = 2 H, "
                       "sW = 2 W, 16-byte aligned output, float32 weights, Bayer)");
        return -3;
    }
    if (lmin && !quad && !x3) {
        hhsr_set_error("hhsr_merge_burst: HHSR_MERGE_LOCAL_MIN needs the x2 or the x3 kernel (scale 2 or 3 on a Bayer sensor, "
                       "ts %% 16 == 0, sH = scale H, sW = scale W, row0 on the tile grid, float32 weights)");
        return -3;
    }
    if (mono && !quad) {  // (tiled is false for monochrome launches unless the x2 conditions hold)
        if (lmin) {
            hhsr_set_error("hhsr_merge_burst: HHSR_MERGE_LOCAL_MIN with HHSR_SENSOR_MONO needs the x2 tile kernel (scale 2, "
                           "ts %% 16 == 0, sH = 2 H, sW = 2 W, row0 %% 32 == 0, float32 weights)");
            return -3;
        }
#define HHSR_MB(WT, GEOM, ISO) hipLaunchKernelGGL((k_merge_burst<WT, GEOM, ISO>), grid, block, 0, s, a, g, c, num, den)
        if (f64) { if (iso) HHSR_MB(double, GEOM_F64, true); else HHSR_MB(double, GEOM_F64, false); }
        else if (p2) { if (iso) HHSR_MB(float, GEOM_P2, true); else HHSR_MB(float, GEOM_P2, false); }
        else { if (iso) HHSR_MB(float, GEOM_F64, true); else HHSR_MB(float, GEOM_F64, false); }
#undef HHSR_MB
    } else if (mono) {  // `mode: grey` at x2: the first-generation tile kernel with a per-pixel covariance window
        hhsr_launch_merge_quad(iso != 0, lmin, true, dim3(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 2, QT)), s, a, g, c, num, den);
    } else if (quad && !x2_v1 && aligned16 && cfa_is_bayer(c)) {  // x2: one wave per parity class
        hhsr_launch_merge_x2(iso != 0, lmin, dim3(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 2, QT)), s, a, g, c, num, den);
    } else if (quad) {  // x2, first generation (non-Bayer 2 x 2 colour layouts, unaligned outputs, HHSR_MERGE_FORCE_X2V1)
        hhsr_launch_merge_quad(iso != 0, lmin, false, dim3(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 2, QT)), s, a, g, c, num, den);
    } else if (x3) {  // x3 (Bayer): the wave-per-parity-class kernel with 3 x 3 sub-pixels per thread
        hhsr_launch_merge_x3(iso != 0, lmin, dim3(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 3, QT)), s, a, g, c, num, den);
    } else if (tiled) {  // other integer scales (and x3 on non-Bayer layouts / with HHSR_MERGE_FORCE_TILE)
        hhsr_launch_merge_tile(p2, iso != 0, dim3(hhsr_cdiv(sW, MT), hhsr_cdiv(nrows, MT)), s, a, g, c, num, den);
    } else {
#define HHSR_MB(WT, GEOM, ISO) hipLaunchKernelGGL((k_merge_burst<WT, GEOM, ISO>), grid, block, 0, s, a, g, c, num, den)
        if (f64) { if (iso) HHSR_MB(double, GEOM_F64, true); else HHSR_MB(double, GEOM_F64, false); }
        else if (p2) { if (iso) HHSR_MB(float, GEOM_P2, true); else HHSR_MB(float, GEOM_P2, false); }
        else { if (iso) HHSR_MB(float, GEOM_F64, true); else HHSR_MB(float, GEOM_F64, false); }
#undef HHSR_MB
    }
    if (!f64 && !(flags & HHSR_MERGE_STORE_CLASSES)) {  // the border bands the float32 kernels skipped, with the reference's float64 weight chain
        const int nf = n_frames + ((flags & HHSR_MERGE_DO_REF) ? 1 : 0);
        static const bool border_v1 = getenv("HHSR_MERGE_BORDER_V1") != nullptr;  // A/B switch, read once
        if (nf >= 2 && nf <= 64 && !border_v1) {  // lane = (pixel, frame)
            const int ppw = 64 / nf;
            const int64_t npx = (int64_t)(g.bt + g.bb) * g.sW + (int64_t)(g.bl + g.br) * (g.sH - g.bt - g.bb);
            const int64_t nwaves = (npx + ppw - 1) / ppw;
            if (npx > 0) {
                const dim3 bgrid((unsigned)((nwaves + 3) / 4)), bblock(256);
                if (iso) hipLaunchKernelGGL((k_merge_border_wave<true>), bgrid, bblock, 0, s, a, g, c, num, den, nf, ppw);
                else hipLaunchKernelGGL((k_merge_border_wave<false>), bgrid, bblock, 0, s, a, g, c, num, den, nf, ppw);
            }
        } else {
            launch_border(g, [&](dim3 bgrid, dim3 bblock) {
                if (iso) hipLaunchKernelGGL((k_merge_border<true>), bgrid, bblock, 0, s, a, g, c, num, den);
                else hipLaunchKernelGGL((k_merge_border<false>), bgrid, bblock, 0, s, a, g, c, num, den);
            });
        }
    }
    HHSR_LAUNCHED();
}


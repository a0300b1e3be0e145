// First-generation LDS tile kernels of the fused burst merge (see hhsr_merge.h for the family overview).
#include "hhsr_merge.h"

template <int GEOM, bool ISO>
__global__ void __launch_bounds__(256) k_merge_burst_tile(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                           float* __restrict__ den) {
    __shared__ float s_raw[RWIN * RPITCH];
    __shared__ float4 s_cov[CWIN * CWIN];
    const int tx = threadIdx.x & (MT - 1), ty = threadIdx.x >> 4;
    // XCD-aware workgroup -> tile mapping: the dispatcher places workgroup b on XCD b % 8 (observed, used for
    // L2 locality only).  Give every XCD one contiguous band of tile rows so that the heavily overlapping
    // windows of neighbouring tiles hit the same 4 MB L2 instead of being fetched once per XCD.
    const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
    int bid = blockIdx.y * nbx + blockIdx.x;
    {   // bijection: XCD x owns ids {b : b % 8 == x} -> contiguous tiles [start_x, start_x + count_x)
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, rem = nblk & 7;
        bid = xcd * q + min(xcd, rem) + loc;
    }
    const int hx0 = (bid % nbx) * MT, hy0 = g.row0 + (bid / nbx) * MT;
    const int hj = hx0 + tx, hi = hy0 + ty;
    const bool live = hj < g.sW && hi < g.row1;
    // corner pixels of the workgroup (clamped into the image) bound every thread's window
    const Pix p0 = make_pix(g, min(hy0, g.row1 - 1), min(hx0, g.sW - 1));
    const Pix p = make_pix(g, min(hi, g.row1 - 1), min(hj, g.sW - 1));
    const int tile = p0.tile;  // uniform: the workgroup lies inside one flow tile
    float n4[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, d4[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float racc = 0.f;  // sum of this pixel's robustness over the frames

    // staging slots of this thread: raw window elements tid and tid+256, covariance element tid.  Only the
    // (MT/s + 3)^2 raw pixels and (MT/(2s) + 3)^2 covariance cells the taps can reach are fetched.
    const int rwin = min(RWIN, (MT + a.iscale - 1) / a.iscale + 3);
    const int cwin = min(CWIN, (MT + 2 * a.iscale - 1) / (2 * a.iscale) + 3);
    const int e0 = threadIdx.x, e1 = threadIdx.x + 256;
    const int e0y = e0 / rwin, e0x = e0 - e0y * rwin, e1y = e1 / rwin, e1x = e1 - e1y * rwin;
    const int cey = threadIdx.x / cwin, cex = threadIdx.x - cey * cwin;
    const bool has0 = e0 < rwin * rwin, has1 = e1 < rwin * rwin, hasc = threadIdx.x < cwin * cwin;

    // GEOM_F64 gives cj = 0 for invalid corners; recompute the corner centre without the validity clamp
    auto corner_centre = [&](const float2 fl, const Pix& pc, int& cj, int& ci, int& x0, int& y0) {
        if (GEOM == GEOM_P2) {
            const FrameGeo q = frame_geom<GEOM, ISO>(fl, g, pc);
            cj = q.cj; ci = q.ci;
        } else {
            cj = (int)floor(pc.lr_x + (double)fl.x);
            ci = (int)floor(pc.lr_y + (double)fl.y);
        }
        x0 = cj >= 1 ? (cj - 1) >> 1 : 0;
        y0 = ci >= 1 ? (ci - 1) >> 1 : 0;
        if (GEOM != GEOM_P2 && !ISO) {  // the float64 path truncates kmap itself; same value for cj >= 1
            x0 = max((int)trunc((pc.lr_x + (double)fl.x) / 2.0 - 0.5), 0);
            y0 = max((int)trunc((pc.lr_y + (double)fl.y) / 2.0 - 0.5), 0);
        }
    };

    float pr0 = 0.f, pr1 = 0.f, plr = 0.f;  // prefetched raw elements and robustness
    float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 pfl = make_float2(0.f, 0.f);
    TileWin pw{0, 0, 0, 0};
    auto prefetch = [&](int n) {
        const FramePtr f = a.f[n];
        pfl = f.flow[tile];
        int cj, ci, x0, y0;
        corner_centre(pfl, p0, cj, ci, x0, y0);
        pw.rx0 = cj - 1; pw.ry0 = ci - 1; pw.cx0 = x0; pw.cy0 = y0;
        if (has0) {
            const int y = pw.ry0 + e0y, x = pw.rx0 + e0x;
            pr0 = (y >= 0 && y < g.H && x >= 0 && x < g.W) ? f.raw[(size_t)y * g.pitch + x] : 0.f;
        }
        if (has1) {
            const int y = pw.ry0 + e1y, x = pw.rx0 + e1x;
            pr1 = (y >= 0 && y < g.H && x >= 0 && x < g.W) ? f.raw[(size_t)y * g.pitch + x] : 0.f;
        }
        if (!ISO && hasc) {
            const int y = min(max(pw.cy0 + cey, 0), g.gh - 1), x = min(max(pw.cx0 + cex, 0), g.gw - 1);
            pc = f.cov[(size_t)y * g.gw + x];
        }
        plr = f.r[p.ridx];
    };

    if (a.n > 0) prefetch(0);
    for (int n = 0; n < a.n; ++n) {
        __syncthreads();  // the previous frame's taps are done with the LDS windows
        if (has0) s_raw[e0y * RPITCH + e0x] = pr0;
        if (has1) s_raw[e1y * RPITCH + e1x] = pr1;
        if (!ISO && hasc) s_cov[cey * CWIN + cex] = pc;
        const float2 fl = pfl;
        const TileWin w = pw;
        const float local_r = plr;
        __syncthreads();
        if (n + 1 < a.n) prefetch(n + 1);  // in flight while this frame's taps are evaluated
        racc += local_r;
        const FrameGeo q = frame_geom<GEOM, ISO>(fl, g, p);
        if (live && q.valid && local_r != 0.f) {
            const float* __restrict__ rc = s_raw + (q.ci - w.ry0) * RPITCH + (q.cj - w.rx0);
            const int lx0 = q.x0 - w.cx0, ly0 = q.y0 - w.cy0;
            const int lx1 = min(q.x0 + 1, g.gw - 1) - w.cx0, ly1 = min(q.y0 + 1, g.gh - 1) - w.cy0;
            taps_accum<ISO, false>(
                q, g, local_r, [=](int di, int dj) { return rc[di * RPITCH + dj]; },
                [=](int k) { return s_cov[(k & 2 ? ly1 : ly0) * CWIN + (k & 1 ? lx1 : lx0)]; }, n4, d4);
        }
    }
    if (!live) return;
    if (owns_lr_pixel(a, hi, hj))
        a.acc_r[p.ridx] = ((a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[p.ridx] : 0.f) + racc;
    if (border_pixel(g, hi, hj)) return;  // k_merge_border's
    const size_t o = ((size_t)(hi - g.row0) * g.sW + hj) * 3;
    float n3[3] = {0.f, 0.f, 0.f}, d3[3] = {0.f, 0.f, 0.f};
    if (a.flags & HHSR_MERGE_LOAD_ACC) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n3[k] = num[o + k];
            d3[k] = den[o + k];
        }
    }
    if (a.flags & HHSR_MERGE_DO_REF) ref_accum_fast<ISO>(a.ref_raw, a.ref_cov, g, hi, hj, n4, d4);
    classes_to_rgb(cfa, n4, d4, n3, d3);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        num[o + k] = (a.flags & HHSR_MERGE_DIVIDE) ? n3[k] / d3[k] : n3[k];
        if (a.flags & HHSR_MERGE_STORE_DEN) den[o + k] = d3[k];
    }
}


template <bool ISO, bool LMIN, bool MONO = false>
__global__ void __launch_bounds__(256) k_merge_burst_quad(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                           float* __restrict__ den) {
    __shared__ float s_raw[RWIN * RPITCH];
    __shared__ float4 s_cov[MONO ? (QT + 3) * CWM : CWIN * CWIN];
    __shared__ float s_R[LMIN ? (QT + 4) * (QT + 4 + 1) : 1];  // LMIN: un-filtered robustness of the tile + 2-pixel border
    quad_tile_body<ISO, LMIN, MONO>(a, g, cfa, num, den, s_raw, s_cov, s_R);
}

void hhsr_launch_merge_tile(bool p2, bool iso, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g, const Cfa4& c,
                            float* num, float* den) {
    const dim3 block(256);
#define HHSR_MT(GEOM, ISO) hipLaunchKernelGGL((k_merge_burst_tile<GEOM, ISO>), grid, block, 0, s, a, g, c, num, den)
    if (p2) { if (iso) HHSR_MT(GEOM_P2, true); else HHSR_MT(GEOM_P2, false); }
    else { if (iso) HHSR_MT(GEOM_F64, true); else HHSR_MT(GEOM_F64, false); }
#undef HHSR_MT
}

void hhsr_launch_merge_quad(bool iso, bool lmin, bool mono, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g,
                            const Cfa4& c, float* num, float* den) {
    const dim3 block(256);
#define HHSR_MQ(ISO, LMIN, MONO) hipLaunchKernelGGL((k_merge_burst_quad<ISO, LMIN, MONO>), grid, block, 0, s, a, g, c, num, den)
    if (mono) {
        if (lmin) { if (iso) HHSR_MQ(true, true, true); else HHSR_MQ(false, true, true); }
        else { if (iso) HHSR_MQ(true, false, true); else HHSR_MQ(false, false, true); }
    } else {
        if (lmin) { if (iso) HHSR_MQ(true, true, false); else HHSR_MQ(false, true, false); }
        else { if (iso) HHSR_MQ(true, false, false); else HHSR_MQ(false, false, false); }
    }
#undef HHSR_MQ
}

// Internal header of the merge translation units (hhsr_merge*.hip): the arithmetic of Alg. 4 / Alg. 11 as device
// functions, the launch arguments and the per-family launchers.  Kernel families, one translation unit each:
//   hhsr_merge.hip        C ABI + kernel choice; per-pixel kernels (k_accumulate, k_accumulate_ref, k_merge_burst: any
//                         scale, float64 validation mode) and the border bands (k_merge_border, _wave)
//   hhsr_merge_tile.hip   first-generation LDS tile kernels: k_merge_burst_tile (integer scales), k_merge_burst_quad (x2;
//                         `mode: grey`)
//   hhsr_merge_x2.hip     k_merge_x2: x2, one wave per Bayer parity class (the headline configuration's kernel)
//   hhsr_merge_xs.hip     k_merge_xs<3>: the same design with S x S sub-pixels per thread (x3)
//
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
#pragma once
#include "hhsr_common.h"
#include <type_traits>

struct Cfa4 {
    uint8_t c[4];
};

struct Geo {
    int H, W, pitch;    // raw frame
    int gh, gw;         // covariance grid (H/2, W/2; monochrome sensors: H, W)
    int mono;           // `mode: grey`: one channel, covariances per pixel (merge.py:131-137, 349-354, 410)
    int ny, nx, ts;     // flow tile grid
    int sH, sW;         // output
    int row0, row1;     // output rows [row0, row1) handled by this launch (merge_burst slabs); num/den point at row0
    int off_lr, off_hr; // sub-images (multi-GPU row slabs): raw row 0 of this image is row off_lr of the full frame and
                        // output row 0 is row off_hr = off_lr * scale: positions are evaluated at FULL-FRAME coordinates
                        // and shifted back exactly, so that their float64 / float32 roundings (the reference keeps the
                        // reference frame's position idx / scale in float32, merge.py:113-114) do not depend on the split
    int bt, bb, bl, br; // border bands: output rows < bt / >= sH - bb and columns < bl / >= sW - br are the pixels whose
                        // reference-frame window centre lies on the outermost raw row / column (see border_pixel)
    double scale;
};

// Border pixels.  A colour can be missing from the reference frame's 3x3 window only when that window is centred on
// the outermost raw row / column; such a pixel's channel sum may then consist of nothing but far-off samples whose
// weights sit at the float32 denormal limit (or below it).  The reference evaluates those weights in float64 and
// rounds the products into float32 accumulators (merge.py:419-434), so 1e-40 / 1e-40 is a colour there and the
// float32 weight chain cannot reproduce it.  The float32 kernels therefore leave the border bands alone and
// k_merge_border computes them with the reference's float64 chain (a few rows / columns: ~0.2 % of the pixels).
__device__ __forceinline__ bool border_pixel(const Geo& g, int hi, int hj) {
    return hi < g.bt || hi >= g.sH - g.bb || hj < g.bl || hj >= g.sW - g.br;
}

// Robustness of the raw pixel (i_r, j_r): the map itself, or (LMIN maps hold the thresholded R) its 5x5
// clamp-border minimum (robustness.py:641-686).
__device__ __forceinline__ float robustness_at(const float* __restrict__ r, const Geo& g, int i_r, int j_r, bool lmin) {
    if (!lmin) return r[(size_t)i_r * g.W + j_r];
    float m = r[(size_t)i_r * g.W + j_r];
    for (int di = -2; di <= 2; ++di) {
        const float* row = r + (size_t)min(max(i_r + di, 0), g.H - 1) * g.W;
        for (int dj = -2; dj <= 2; ++dj) m = fminf(m, row[min(max(j_r + dj, 0), g.W - 1)]);
    }
    return m;
}

struct FramePtr {
    const float* raw;
    const float2* flow;
    const float4* cov;
    const float* r;
};

// ---- one comp frame's contribution to HR pixel (hi, hj)  (merge.py:291-434) -----------------------
template <typename WT, bool ISO>
__device__ __forceinline__ void comp_contrib(const FramePtr f, const Geo& g, const Cfa4 cfa, int hi, int hj,
                                             float val[3], float acc[3], bool lmin = false) {
    const double lr_x = ((double)hj + 0.5) / g.scale;
    const double lr_y = ((double)(hi + g.off_hr) + 0.5) / g.scale - (double)g.off_lr;
    const int px = (int)lr_x / g.ts, py = (int)lr_y / g.ts;  // == int(lr // tile_size) for lr >= 0
    const float2 fl = f.flow[(size_t)py * g.nx + px];
    const int i_r = min((int)lr_y, g.H - 1), j_r = min((int)lr_x, g.W - 1);
    const double mx = lr_x + (double)fl.x, my = lr_y + (double)fl.y;
    if (!(mx >= 0.0 && mx < (double)g.W && my >= 0.0 && my < (double)g.H)) return;
    const WT local_r = (WT)robustness_at(f.r, g, i_r, j_r, lmin);
    WT ixx = 0, ixy = 0, iyy = 0;
    if (!ISO) {
        const double kj = g.mono ? mx - 0.5 : mx / 2.0 - 0.5, ki = g.mono ? my - 0.5 : my / 2.0 - 0.5;
        const double tkj = trunc(kj), tki = trunc(ki);
        const WT fx = (WT)(kj - tkj), fy = (WT)(ki - tki);  // signed modf fraction (D11)
        const int x0 = max((int)tkj, 0), y0 = max((int)tki, 0);
        const int x1 = min(x0 + 1, g.gw - 1), y1 = min(y0 + 1, g.gh - 1);
        const float4 c00 = f.cov[(size_t)y0 * g.gw + x0], c01 = f.cov[(size_t)y0 * g.gw + x1];
        const float4 c10 = f.cov[(size_t)y1 * g.gw + x0], c11 = f.cov[(size_t)y1 * g.gw + x1];
        // float32 differences, then lerp in the weight type (merge.py:378-390)
        const WT txx = (WT)c00.x + fx * (WT)(c01.x - c00.x), bxx = (WT)c10.x + fx * (WT)(c11.x - c10.x);
        const WT txy = (WT)c00.y + fx * (WT)(c01.y - c00.y), bxy = (WT)c10.y + fx * (WT)(c11.y - c10.y);
        const WT tyy = (WT)c00.w + fx * (WT)(c01.w - c00.w), byy = (WT)c10.w + fx * (WT)(c11.w - c10.w);
        const WT cxx = txx + fy * (bxx - txx), cxy = txy + fy * (bxy - txy), cyy = tyy + fy * (byy - tyy);
        const WT det = cxx * cyy - cxy * cxy;
        const WT inv_det = (WT)1.0 / det;
        ixx = inv_det * cyy;
        ixy = -inv_det * cxy;
        iyy = inv_det * cxx;
    }
    const int cj = (int)mx, ci = (int)my;
    const double mj = mx - 0.5, mi = my - 0.5;
#pragma unroll
    for (int di = -1; di <= 1; ++di) {
        const int i = ci + di;
        const WT dy = (WT)((double)i - mi);
#pragma unroll
        for (int dj = -1; dj <= 1; ++dj) {
            const int j = cj + dj;
            if (j < 0 || j >= g.W || i < 0 || i >= g.H) continue;
            const int ch = cfa.c[(i & 1) * 2 + (j & 1)];  // (monochrome: the host passes an all-zero pattern)
            const WT c = (WT)f.raw[(size_t)i * g.pitch + j];
            const WT dx = (WT)((double)j - mj);
            WT z;
            if (ISO) z = (WT)2.0 * (dx * dx + dy * dy);
            else z = ixx * dx * dx + (WT)2.0 * ixy * dx * dy + iyy * dy * dy;
            z = z > (WT)0 ? z : (WT)0;  // Python max(0, z): NaN -> 0 -> w = 1 (D10)
            const WT w = exp((WT)-0.5 * z);
            const WT wr = w * local_r;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (ch == k) {
                    val[k] = (float)((WT)val[k] + wr * c);
                    acc[k] = (float)((WT)acc[k] + wr);
                }
        }
    }
}

// ---- float32 fast path of the same contribution ------------------------------------------------------
// The discrete decisions (which LR pixel is the window centre, which covariance cell, in/out of frame)
// are taken exactly as the reference's float64 code takes them; everything continuous — covariance
// interpolation / inversion, the quadratic form, exp — runs in float32:
//   * z = (ixx*dx + 2 ixy*dy)*dx + iyy*dy*dy: two FMAs per tap with the per-row terms hoisted;
//   * w = exp(-z/2) = v_exp_f32(z * -0.5*log2(e))  (abs error < 1e-7 on weights in [0, 1]);
//   * the CFA channel of a tap depends only on the parity of its raw coordinates, so the 9 taps are
//     summed into 4 parity-class accumulators with compile-time indices and the classes are mapped to
//     R/G/B once per OUTPUT PIXEL (the CFA is wave-uniform) instead of a 3-way select per tap;
//   * frames whose robustness is exactly 0 at this pixel add +0 to both sums and are skipped.
// Two geometry front ends:
//   GEOM_P2   scale in {1, 2, 4, ...}: (hj + 0.5)/scale and its split into integer + fraction are exact
//             in float32, and floor(lr + flow) is decided by ONE exact float comparison
//             flow >= floor(flow) + (1 - frac(lr)) — no float64 instruction in the frame loop;
//   GEOM_F64  any scale: positions in float64 exactly like the reference (merge.py:319-345, 396-399).
//             (An exact integer + float32 form for odd integer scales — h = s q + rem, carry when frac(flow) >=
//             (2 s - 2 rem - 1)/(2 s) — was built and measured in round 2: x3 at 48 MP 45.2 ms vs 44.4 ms with this
//             float64 geometry, i.e. no gain: the tile kernel is bound by the ~250 VALU instructions of per-pixel
//             covariance blend + 9 taps, not by its ~25 float64 operations.  Note for a retry: frac(flow) =
//             flow - floor(flow) is NOT exact in float32 for small negative flows (-0.1 + 1 rounds), so the carry
//             must be decided as flow >= floor(flow) + t like GEOM_P2 does, or in float64.)
// Differences to the float64 weight chain are O(1e-6) relative on num/den (tests: rel 2e-5).
enum { GEOM_F64 = 0, GEOM_P2 = 1 };

struct Pix {
    // frame-independent per-output-pixel state
    double lr_x, lr_y;   // GEOM_F64
    int lix, liy;        // GEOM_P2: integer part of the LR position ...
    float lfx, lfy;      // ... and its exact fraction
    int tile;            // flow tile index
    int ridx;            // robustness pixel index
};

__device__ __forceinline__ Pix make_pix(const Geo& g, int hi, int hj) {
    Pix p;
    p.lr_x = ((double)hj + 0.5) / g.scale;
    p.lr_y = ((double)(hi + g.off_hr) + 0.5) / g.scale - (double)g.off_lr;
    p.lix = (int)p.lr_x;
    p.liy = (int)p.lr_y;
    p.lfx = (float)(p.lr_x - (double)p.lix);
    p.lfy = (float)(p.lr_y - (double)p.liy);
    p.tile = (p.liy / g.ts) * g.nx + p.lix / g.ts;
    p.ridx = min(p.liy, g.H - 1) * g.W + min(p.lix, g.W - 1);
    return p;
}

// Per-frame geometry of one output pixel: window centre, fractions, covariance cell.
struct FrameGeo {
    int ci, cj;      // centre raw pixel = int(lr + flow)
    int x0, y0;      // top-left covariance cell
    float frx, fry;  // lr + flow - centre, in [0, 1)
    float fx, fy;    // signed fraction of the covariance position (D11)
    bool valid;      // lr + flow inside the frame
};

template <int GEOM, bool ISO>
__device__ __forceinline__ FrameGeo frame_geom(const float2 fl, const Geo& g, const Pix& p) {
    FrameGeo q;
    q.x0 = q.y0 = 0;
    q.fx = q.fy = 0.f;
    if (GEOM == GEOM_P2) {
        const float fix = floorf(fl.x), fiy = floorf(fl.y);
        const int cx = fl.x >= fix + (1.f - p.lfx), cy = fl.y >= fiy + (1.f - p.lfy);  // exact
        q.cj = p.lix + (int)fix + cx;
        q.ci = p.liy + (int)fiy + cy;
        q.valid = q.cj >= 0 && q.cj < g.W && q.ci >= 0 && q.ci < g.H;
        q.frx = (fl.x - fix) + (p.lfx - (float)cx);
        q.fry = (fl.y - fiy) + (p.lfy - (float)cy);
        if (!ISO && !g.mono) {  // kmap = lr_mov/2 - 0.5, trunc toward zero + signed fraction (merge.py:349-361)
            if (q.cj >= 1) { q.x0 = (q.cj - 1) >> 1; q.fx = 0.5f * ((float)((q.cj - 1) & 1) + q.frx); }
            else           { q.x0 = 0;               q.fx = 0.5f * (q.frx - 1.f); }
            if (q.ci >= 1) { q.y0 = (q.ci - 1) >> 1; q.fy = 0.5f * ((float)((q.ci - 1) & 1) + q.fry); }
            else           { q.y0 = 0;               q.fy = 0.5f * (q.fry - 1.f); }
        } else if (!ISO) {      // monochrome: kmap = lr_mov - 0.5 = c + fr - 0.5 (fr +- 0.5 is exact)
            const bool hx = q.frx >= 0.5f, hy = q.fry >= 0.5f;
            q.x0 = hx ? q.cj : max(q.cj - 1, 0);
            q.fx = hx ? q.frx - 0.5f : (q.cj >= 1 ? q.frx + 0.5f : q.frx - 0.5f);
            q.y0 = hy ? q.ci : max(q.ci - 1, 0);
            q.fy = hy ? q.fry - 0.5f : (q.ci >= 1 ? q.fry + 0.5f : q.fry - 0.5f);
        }
    } else {
        const double mx = p.lr_x + (double)fl.x, my = p.lr_y + (double)fl.y;
        q.valid = mx >= 0.0 && mx < (double)g.W && my >= 0.0 && my < (double)g.H;
        q.cj = q.valid ? (int)mx : 0;
        q.ci = q.valid ? (int)my : 0;
        q.frx = (float)(mx - (double)q.cj);
        q.fry = (float)(my - (double)q.ci);
        if (!ISO) {
            const double kj = g.mono ? mx - 0.5 : mx / 2.0 - 0.5, ki = g.mono ? my - 0.5 : my / 2.0 - 0.5;
            const double tkj = trunc(kj), tki = trunc(ki);
            q.fx = (float)(kj - tkj);
            q.fy = (float)(ki - tki);
            q.x0 = q.valid ? max((int)tkj, 0) : 0;
            q.y0 = q.valid ? max((int)tki, 0) : 0;
        }
    }
    return q;
}

// The 9 taps of one frame -> absolute-parity accumulators n4/d4[row parity][col parity].
// rawAt(di, dj): raw sample at (ci+di, cj+dj); covAt(k): covariance of cell k = (y0|y1, x0|x1).
// REF = the reference frame's variant (merge.py:83-233): the inverse falls back to the identity when
// |det| <= 1e-10 or NaN (linalg.py:53-64) instead of propagating NaN.
template <bool ISO, bool REF, class RawAt, class CovAt>
__device__ __forceinline__ void taps_accum(const FrameGeo& q, const Geo& g, const float local_r, RawAt rawAt,
                                           CovAt covAt, float n4[2][2], float d4[2][2]) {
    float ixx = 2.f, ixy = 0.f, iyy = 2.f;  // iso kernel: z = 2 (dx^2 + dy^2)
    if (!ISO) {
        const float4 c00 = covAt(0), c01 = covAt(1), c10 = covAt(2), c11 = covAt(3);
        const float fx = q.fx, fy = q.fy;
        const float txx = c00.x + fx * (c01.x - c00.x), bxx = c10.x + fx * (c11.x - c10.x);
        const float txy = c00.y + fx * (c01.y - c00.y), bxy = c10.y + fx * (c11.y - c10.y);
        const float tyy = c00.w + fx * (c01.w - c00.w), byy = c10.w + fx * (c11.w - c10.w);
        const float cxx = txx + fy * (bxx - txx), cxy = txy + fy * (bxy - txy), cyy = tyy + fy * (byy - tyy);
        const float det = cxx * cyy - cxy * cxy;
        const float inv_det = __builtin_amdgcn_rcpf(det);
        ixx = inv_det * cyy;
        ixy = -inv_det * cxy;
        iyy = inv_det * cxx;
        if (REF && !(fabsf(det) > 1e-10f)) {
            ixx = 1.f;
            ixy = 0.f;
            iyy = 1.f;
        }
    }
    const float dx0 = 0.5f - q.frx, dy0 = 0.5f - q.fry;  // tap - (lr_mov - 0.5) for the centre tap
    // w = exp(-z/2) = exp2(z * kexp), kexp = -0.5*log2(e) folded into the quadratic form; since kexp < 0
    // the clamp max(0, z) becomes min(0, kexp*z), which also maps NaN -> 0 -> w = 1 (D10).  The robustness
    // factor is applied once to the four class sums instead of to every tap weight.
    // v_exp_f32 flushes results below 2^-126 to zero, but the reference keeps weights down to the float32
    // denormal limit in its float32 accumulators (a far-off sample can be the ONLY sample of a colour in a
    // border pixel's window: 1e-40/1e-40 is a colour, 0/0 is NaN).  So e = exp2(z/2) is evaluated with the
    // hardware instruction (normal down to z = -252) and w = e*e underflows gradually (f32 denormals are on).
    const float kexp = -0.36067376022224085184f;  // -0.25 * log2(e): exp(-q/2) = (exp2(q * kexp))^2
    ixx *= kexp;
    ixy *= 2.f * kexp;
    iyy *= kexp;
    float sv[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, sa[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // by OFFSET parity
    const int ci = q.ci, cj = q.cj;
    const bool interior = ci >= 1 && ci + 1 < g.H && cj >= 1 && cj + 1 < g.W;
    const float dxm = dx0 - 1.f, dxp = dx0 + 1.f;
    // the 9 taps; CHECK = window crosses the frame border (taps outside are skipped, merge.py:404-405).  Two
    // copies so that the common interior case is straight-line code without per-tap exec-mask branches.
    auto taps = [&](auto check) {
        constexpr bool CHECK = decltype(check)::value;
#pragma unroll
        for (int di = -1; di <= 1; ++di) {
            const float dy = dy0 + (float)di;
            const float a = iyy * dy * dy, b = ixy * dy;
#pragma unroll
            for (int dj = -1; dj <= 1; ++dj) {
                if (CHECK && (cj + dj < 0 || cj + dj >= g.W || ci + di < 0 || ci + di >= g.H)) continue;
                const float c = rawAt(di, dj);
                const float dx = dj < 0 ? dxm : (dj > 0 ? dxp : dx0);
                const float z = fminf(fmaf(fmaf(ixx, dx, b), dx, a), 0.f);
                const float e = __builtin_amdgcn_exp2f(z);
                const float w = e * e;
                sv[di & 1][dj & 1] = fmaf(w, c, sv[di & 1][dj & 1]);
                sa[di & 1][dj & 1] += w;
            }
        }
    };
    if (interior) taps(std::false_type{});
    else taps(std::true_type{});
    // offset parity -> absolute raw-coordinate parity: swap columns / rows when the centre is odd
    const bool oj = cj & 1, oi = ci & 1;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float v0 = oj ? sv[r][1] : sv[r][0], v1 = oj ? sv[r][0] : sv[r][1];
        const float a0 = oj ? sa[r][1] : sa[r][0], a1 = oj ? sa[r][0] : sa[r][1];
        sv[r][0] = v0; sv[r][1] = v1; sa[r][0] = a0; sa[r][1] = a1;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        n4[0][c] = fmaf(local_r, oi ? sv[1][c] : sv[0][c], n4[0][c]);
        n4[1][c] = fmaf(local_r, oi ? sv[0][c] : sv[1][c], n4[1][c]);
        d4[0][c] = fmaf(local_r, oi ? sa[1][c] : sa[0][c], d4[0][c]);
        d4[1][c] = fmaf(local_r, oi ? sa[0][c] : sa[1][c], d4[1][c]);
    }
}

// One comp frame, operands straight from global memory (generic scales / per-frame operator API).
template <int GEOM, bool ISO>
__device__ __forceinline__ void comp_accum_fast(const FramePtr f, const Geo& g, const Pix& p, float n4[2][2],
                                                float d4[2][2], bool lmin = false) {
    const FrameGeo q = frame_geom<GEOM, ISO>(f.flow[p.tile], g, p);
    if (!q.valid) return;
    const float local_r = lmin ? robustness_at(f.r, g, p.ridx / g.W, p.ridx % g.W, true) : f.r[p.ridx];
    if (local_r == 0.f) return;
    const float* __restrict__ rawc = f.raw + (size_t)q.ci * g.pitch + q.cj;
    const int x1 = min(q.x0 + 1, g.gw - 1), y1 = min(q.y0 + 1, g.gh - 1);
    const float4* __restrict__ r0 = ISO ? nullptr : f.cov + (size_t)q.y0 * g.gw;
    const float4* __restrict__ r1 = ISO ? nullptr : f.cov + (size_t)y1 * g.gw;
    const int x0 = q.x0, pitch = g.pitch;
    taps_accum<ISO, false>(
        q, g, local_r, [=](int di, int dj) { return rawc[di * pitch + dj]; },
        [=](int k) { return (k & 2 ? r1 : r0)[k & 1 ? x1 : x0]; }, n4, d4);
}

// Reference frame, float32 weights, no accumulated-robustness denoiser (merge.py:83-233 with rad = 1).
// Position = idx/scale stored in float32 like the reference's local array; centre = round-half-even;
// covariance cell from floor((pos - 0.5)/2) with the signed modf fraction (linalg.py:190-200).
template <bool ISO>
__device__ __forceinline__ void ref_accum_fast(const float* __restrict__ raw, const float4* __restrict__ cov,
                                               const Geo& g, int oi, int oj, float n4[2][2], float d4[2][2]) {
    const float pyf = (float)((double)(oi + g.off_hr) / g.scale) - (float)g.off_lr, pxf = (float)((double)oj / g.scale);
    FrameGeo q;
    q.cj = (int)rintf(pxf);
    q.ci = (int)rintf(pyf);
    q.frx = 0.5f - ((float)q.cj - pxf);  // so that dx0 = centre - pos (no half-pixel offset here, D7)
    q.fry = 0.5f - ((float)q.ci - pyf);
    q.valid = true;
    q.x0 = q.y0 = 0;
    q.fx = q.fy = 0.f;
    int x1 = 0, y1 = 0;
    if (!ISO) {
        const float gy = g.mono ? pyf : (pyf - 0.5f) * 0.5f;  // == float32((pos - 0.5)/2); monochrome: the position itself
        const float gx = g.mono ? pxf : (pxf - 0.5f) * 0.5f;
        q.x0 = (int)fmaxf(floorf(gx), 0.f);
        q.y0 = (int)fmaxf(floorf(gy), 0.f);
        q.fx = gx - truncf(gx);
        q.fy = gy - truncf(gy);
        x1 = min(q.x0 + 1, g.gw - 1);
        y1 = min(q.y0 + 1, g.gh - 1);
    }
    const float* __restrict__ rawc = raw + (size_t)q.ci * g.pitch + q.cj;
    const float4* __restrict__ r0 = ISO ? nullptr : cov + (size_t)q.y0 * g.gw;
    const float4* __restrict__ r1 = ISO ? nullptr : cov + (size_t)y1 * g.gw;
    const int x0 = q.x0, pitch = g.pitch;
    taps_accum<ISO, true>(
        q, g, 1.0f, [=](int di, int dj) { return rawc[di * pitch + dj]; },
        [=](int k) { return (k & 2 ? r1 : r0)[k & 1 ? x1 : x0]; }, n4, d4);
}

// parity classes -> channels (wave-uniform CFA): val[cfa[i][j]] += n4[i][j] in fixed order
__device__ __forceinline__ void classes_to_rgb(const Cfa4 cfa, const float n4[2][2], const float d4[2][2],
                                               float val[3], float acc[3]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = cfa.c[i * 2 + j];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (ch == k) {
                    val[k] += n4[i][j];
                    acc[k] += d4[i][j];
                }
        }
}

// ---- the reference frame's contribution (merge.py:83-233) -------------------------------------------
// Returns true when the accumulated-robustness rule says "overwrite" (merge.py:223-228).
template <bool ISO>
__device__ __forceinline__ bool ref_contrib(const float* __restrict__ raw, const float4* __restrict__ cov,
                                            const Geo& g, const Cfa4 cfa, int oi, int oj,
                                            const float* __restrict__ acc_rob, int rad_max, double max_mult,
                                            double max_fc, float val[3], float acc[3]) {
    const float pyf = (float)((double)(oi + g.off_hr) / g.scale) - (float)g.off_lr;  // coarse_ref_sub_pos is a float32 local array
    const float pxf = (float)((double)oj / g.scale);
    float i00 = 1.f, i01 = 0.f, i10 = 0.f, i11 = 1.f;
    if (!ISO) {
        const float gy = g.mono ? pyf : (float)(((double)pyf - 0.5) / 2.0);
        const float gx = g.mono ? pxf : (float)(((double)pxf - 0.5) / 2.0);
        const int x0 = (int)fmaxf(floorf(gx), 0.f), y0 = (int)fmaxf(floorf(gy), 0.f);
        const int x1 = min(x0 + 1, g.gw - 1), y1 = min(y0 + 1, g.gh - 1);
        const double rx = (double)(gx - truncf(gx)), ry = (double)(gy - truncf(gy));  // modf (signed)
        const float4 c00 = cov[(size_t)y0 * g.gw + x0], c01 = cov[(size_t)y0 * g.gw + x1];
        const float4 c10 = cov[(size_t)y1 * g.gw + x0], c11 = cov[(size_t)y1 * g.gw + x1];
#define HHSR_ICOV(m) \
    (float)((double)c00.m * (1.0 - rx) * (1.0 - ry) + (double)c01.m * rx * (1.0 - ry) + \
            (double)c10.m * (1.0 - rx) * ry + (double)c11.m * rx * ry)
        const float m00 = HHSR_ICOV(x), m01 = HHSR_ICOV(y), m10 = HHSR_ICOV(z), m11 = HHSR_ICOV(w);
#undef HHSR_ICOV
        const float det = m00 * m11 - m01 * m10;  // float32 (linalg.py:53)
        if (fabsf(det) > 1e-10f) {                // NaN fails the test -> identity (D10)
            const double det_i = 1.0 / (double)det;
            i00 = (float)((double)m11 * det_i);
            i01 = (float)(-(double)m01 * det_i);
            i10 = (float)(-(double)m10 * det_i);
            i11 = (float)((double)m00 * det_i);
        }
    }
    double power = 1.0;
    int rad = 1;
    bool overwrite = false;
    if (acc_rob) {
        const int ry_i = min((int)rintf(pyf), g.H - 1), rx_i = min((int)rintf(pxf), g.W - 1);
        const float la = acc_rob[(size_t)ry_i * g.W + rx_i];
        if ((double)la <= max_fc) {  // utils_image.py:311-325
            power = max_mult;
            rad = rad_max;
        }
        overwrite = (double)la < max_fc;
    }
    const int cx = (int)rintf(pxf), cy = (int)rintf(pyf);  // round-half-even
    for (int i = -rad; i <= rad; ++i) {
        const int pi = cy + i;
        const double dy = (double)pi - (double)pyf;
        for (int j = -rad; j <= rad; ++j) {
            const int pj = cx + j;
            if (pj < 0 || pj >= g.W || pi < 0 || pi >= g.H) continue;
            const int ch = cfa.c[(pi & 1) * 2 + (pj & 1)];
            const double c = (double)raw[(size_t)pi * g.pitch + pj];
            const double dx = (double)pj - (double)pxf;
            double y;
            if (ISO) y = 2.0 * (dx * dx + dy * dy);
            else y = (double)i00 * dx * dx + dx * dy * (double)(i01 + i10) + (double)i11 * dy * dy;
            y = pymax0(y);
            y = y / power;
            const double w = exp(-0.5 * y);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (ch == k) {
                    val[k] = (float)((double)val[k] + c * w);
                    acc[k] = (float)((double)acc[k] + w);
                }
        }
    }
    return overwrite;
}

// ---- fused burst kernel -------------------------------------------------------------------------------
struct BurstArgs {
    FramePtr f[HHSR_MAX_FRAMES];
    int n;
    const float* ref_raw;
    const float4* ref_cov;
    int flags;
    float* acc_r;  // optional [H][W]: sum of the frames' robustness (integer scales only)
    int iscale;    // (int)scale (integer scales: accumulated robustness ownership, tile window sizes)
    float* cls;    // chained x2 launches (HHSR_MERGE_STORE_CLASSES / _LOAD_CLASSES): per tile 33 x 256 floats
    int first;     // HHSR_MERGE_LOAD_CLASSES: frames [0, first) are already in `cls` for the wave-uniform tiles
};

// The HR pixels with hi % s == 0 and hj % s == 0 map one-to-one onto the LR pixels (integer scale s): they
// carry the accumulated robustness sum_n r_n of "their" LR pixel (super_resolution.py:158-159), which costs
// no extra HBM traffic here because r is read for the merge anyway.
__device__ __forceinline__ bool owns_lr_pixel(const BurstArgs& a, int hi, int hj) {
    return a.acc_r != nullptr && (hi % a.iscale) == 0 && (hj % a.iscale) == 0;
}

// All frames + reference frame + normalisation of ONE output pixel, operands from global memory.
template <typename WT, int GEOM, bool ISO>
__device__ __forceinline__ void merge_pixel(const BurstArgs& a, const Geo& g, const Cfa4 cfa, int hi, int hj,
                                            float* __restrict__ num, float* __restrict__ den) {
    const size_t o = ((size_t)(hi - g.row0) * g.sW + hj) * 3;
    const bool lmin = (a.flags & HHSR_MERGE_LOCAL_MIN) != 0;
    float n3[3] = {0.f, 0.f, 0.f}, d3[3] = {0.f, 0.f, 0.f};
    if (a.flags & HHSR_MERGE_LOAD_ACC) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n3[k] = num[o + k];
            d3[k] = den[o + k];
        }
    }
    if (sizeof(WT) == 4) {
        // fast path: parity-class sums over all frames, mapped to R/G/B once
        const Pix p = make_pix(g, hi, hj);
        float n4[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, d4[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        for (int n = 0; n < a.n; ++n) comp_accum_fast<GEOM, ISO>(a.f[n], g, p, n4, d4, lmin);
        classes_to_rgb(cfa, n4, d4, n3, d3);
    } else {
        for (int n = 0; n < a.n; ++n) {
            float val[3] = {0.f, 0.f, 0.f}, acc[3] = {0.f, 0.f, 0.f};
            comp_contrib<WT, ISO>(a.f[n], g, cfa, hi, hj, val, acc, lmin);
#pragma unroll
            for (int k = 0; k < 3; ++k) {  // same float32 order as successive `num += val`
                n3[k] += val[k];
                d3[k] += acc[k];
            }
        }
    }
    if (a.flags & HHSR_MERGE_DO_REF) {
        float val[3] = {0.f, 0.f, 0.f}, acc[3] = {0.f, 0.f, 0.f};
        ref_contrib<ISO>(a.ref_raw, a.ref_cov, g, cfa, hi, hj, nullptr, 0, 0.0, 0.0, val, acc);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n3[k] += val[k];
            d3[k] += acc[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        num[o + k] = (a.flags & HHSR_MERGE_DIVIDE) ? n3[k] / d3[k] : n3[k];
        if (a.flags & HHSR_MERGE_STORE_DEN) den[o + k] = d3[k];
    }
}


// ---- fused burst kernel with LDS staging per flow tile ------------------------------------------------
// For integer scales the HR tile of one flow vector is ts*scale pixels wide (a multiple of 16), so a 16x16
// HR workgroup aligned to 16 sees ONE flow vector per frame.  Its raw footprint (<= 19x19 pixels) and
// covariance footprint (<= 12x12 cells) are fetched once per frame with coalesced loads, staged in LDS
// and read from there by the 9 taps / 4 covariance cells of every pixel: ~3 vector loads per
// pixel-frame instead of 15 (the un-staged kernel is bound by the L1 request rate, profiles/r01_b).
// Loads for frame n+1 are issued into registers before the taps of frame n are evaluated.
constexpr int MT = 16;                  // HR workgroup edge
constexpr int RWIN = 19, RPITCH = 21;   // raw window: (MT/s + 3) <= 19, odd-ish pitch against bank conflicts
constexpr int CWIN = 12;                // covariance window edge (<= MT/2 + 3 cells)

struct TileWin {
    int rx0, ry0;  // raw window origin (may be negative: outside -> 0, never read by in-frame taps)
    int cx0, cy0;  // covariance window origin
};


// ---- x2 variant: one thread = one LR pixel = its 2 x 2 HR pixels ----------------------------------------------
// At scale 2 the four HR pixels of an LR pixel share the robustness sample, the flow vector, the staged windows
// and most of the geometry (their centres differ by at most one raw pixel, decided by wave-uniform comparisons
// of frac(flow) with 0.25 / 0.75).  A 16 x 16 LR workgroup (32 x 32 HR, inside one flow tile for ts % 16 == 0)
// stages a 19 x 19 raw window and an 11 x 11 covariance window per frame — the per-frame staging, prefetch
// address arithmetic and the two workgroup barriers are paid once per FOUR output pixels, and every thread owns
// exactly one accumulated-robustness sample.  Same arithmetic per HR pixel as k_merge_burst_tile (frame_geom /
// taps_accum are shared), so results are bit-identical to it.
constexpr int QT = 16;  // LR workgroup edge

// Body of the first-generation x2 kernel: general per-pixel geometry, any window position.  s_raw: >= RWIN * RPITCH
// floats, s_cov: >= CWIN * CWIN float4, s_R: >= (QT + 4) * (QT + 5) floats (LMIN).
// MONO (`mode: grey`, merge.py:349-354): one covariance per PIXEL, so the staged covariance window has the raw
// window's extent (19 x 19 cells at pitch CWM) instead of the Bayer grid's 11 x 11; the all-zero CFA pattern of a
// monochrome launch routes the four parity classes into channel 0 (classes_to_rgb).
constexpr int CWM = 20;  // covariance window pitch of the monochrome variant (float4 cells)
template <bool ISO, bool LMIN, bool MONO = false>
__device__ __forceinline__ void quad_tile_body(const BurstArgs& a, const Geo& g, const Cfa4 cfa, float* __restrict__ num,
                                               float* __restrict__ den, float* __restrict__ s_raw,
                                               float4* __restrict__ s_cov, float* __restrict__ s_Rf) {
    float (*s_R)[QT + 4 + 1] = reinterpret_cast<float (*)[QT + 4 + 1]>(s_Rf);
    const int tx = threadIdx.x & (QT - 1), ty = threadIdx.x >> 4;
    const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
    int bid = blockIdx.y * nbx + blockIdx.x;
    {   // XCD-aware tile order, see k_merge_burst_tile
        const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, rem = nblk & 7;
        bid = xcd * q + min(xcd, rem) + loc;
    }
    const int lx0 = (bid % nbx) * QT, ly0 = (g.row0 >> 1) + (bid / nbx) * QT;  // LR origin of the workgroup
    const int lx = lx0 + tx, ly = ly0 + ty;
    const bool live = lx < g.W && 2 * ly < g.row1;
    const int lxc = min(lx, g.W - 1), lyc = min(ly, (g.row1 >> 1) - 1);
    const Pix p0 = make_pix(g, 2 * min(ly0, (g.row1 >> 1) - 1), 2 * min(lx0, g.W - 1));  // smallest centre of the tile
    Pix pq[2][2];
#pragma unroll
    for (int sa = 0; sa < 2; ++sa)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) pq[sa][sb] = make_pix(g, 2 * lyc + sa, 2 * lxc + sb);
    const int tile = p0.tile, ridx = pq[0][0].ridx;
    float n4[2][2][2][2], d4[2][2][2][2];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        (&n4[0][0][0][0])[k] = 0.f;
        (&d4[0][0][0][0])[k] = 0.f;
    }
    float racc = 0.f;

    constexpr int rwin = QT + 3, cwin = MONO ? QT + 3 : QT / 2 + 3;  // 19 raw pixels, 11 (monochrome: 19) covariance cells
    constexpr int CP = MONO ? CWM : CWIN;                             // pitch of the staged covariance window
    static_assert(rwin <= RWIN && cwin <= CP, "window buffers");
    const int e0 = threadIdx.x, e1 = threadIdx.x + 256;
    const int e0y = e0 / rwin, e0x = e0 - e0y * rwin, e1y = e1 / rwin, e1x = e1 - e1y * rwin;
    const int cey = threadIdx.x / cwin, cex = threadIdx.x - cey * cwin;
    const int ce1 = threadIdx.x + 256, ce1y = ce1 / cwin, ce1x = ce1 - ce1y * cwin;  // (monochrome: 361 cells)
    const bool has1 = e1 < rwin * rwin, hasc = threadIdx.x < cwin * cwin, hasc1 = MONO && ce1 < cwin * cwin;

    // LMIN: the frames carry the thresholded map R; r = its 5x5 clamp-border minimum (robustness.py:641-686) is
    // taken here from a (QT+4)^2 window — the separate local-minimum pass and its 8 B/pixel disappear
    constexpr int RW = QT + 4;
    const int m0y = threadIdx.x / RW, m0x = threadIdx.x - m0y * RW;
    const int m1 = threadIdx.x + 256, m1y = m1 / RW, m1x = m1 - m1y * RW;
    const bool hasm1 = LMIN && m1 < RW * RW;
    const int moff0 = clampi(ly0 - 2 + m0y, 0, g.H - 1) * g.W + clampi(lx0 - 2 + m0x, 0, g.W - 1);
    const int moff1 = clampi(ly0 - 2 + m1y, 0, g.H - 1) * g.W + clampi(lx0 - 2 + m1x, 0, g.W - 1);
    float pr0 = 0.f, pr1 = 0.f, plr = 0.f, plr1 = 0.f;
    float4 pc = make_float4(0.f, 0.f, 0.f, 0.f), pc1 = pc;
    float2 pfl = make_float2(0.f, 0.f);
    TileWin pw{0, 0, 0, 0};
    auto prefetch = [&](int n) {
        const FramePtr f = a.f[n];
        pfl = f.flow[tile];
        const FrameGeo qc = frame_geom<GEOM_P2, ISO>(pfl, g, p0);
        pw.rx0 = qc.cj - 1; pw.ry0 = qc.ci - 1;
        if (MONO) {  // cells c - 1 .. c + 1 of every centre c of the tile (frame_geom's monochrome branch)
            pw.cx0 = max(qc.cj - 1, 0);
            pw.cy0 = max(qc.ci - 1, 0);
        } else {
            pw.cx0 = qc.cj >= 1 ? (qc.cj - 1) >> 1 : 0;
            pw.cy0 = qc.ci >= 1 ? (qc.ci - 1) >> 1 : 0;
        }
        {
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
        if (!ISO && hasc1) {
            const int y = min(max(pw.cy0 + ce1y, 0), g.gh - 1), x = min(max(pw.cx0 + ce1x, 0), g.gw - 1);
            pc1 = f.cov[(size_t)y * g.gw + x];
        }
        if (LMIN) {
            plr = f.r[moff0];
            if (hasm1) plr1 = f.r[moff1];
        } else {
            plr = f.r[ridx];
        }
    };

    if (a.n > 0) prefetch(0);
    for (int n = 0; n < a.n; ++n) {
        __syncthreads();  // the previous frame's taps are done with the LDS windows
        s_raw[e0y * RPITCH + e0x] = pr0;
        if (has1) s_raw[e1y * RPITCH + e1x] = pr1;
        if (!ISO && hasc) s_cov[cey * CP + cex] = pc;
        if (!ISO && hasc1) s_cov[ce1y * CP + ce1x] = pc1;
        if (LMIN) {
            s_R[m0y][m0x] = plr;
            if (hasm1) s_R[m1y][m1x] = plr1;
        }
        const float2 fl = pfl;
        const TileWin w = pw;
        float local_r = plr;
        __syncthreads();
        if (n + 1 < a.n) prefetch(n + 1);  // in flight while this frame's taps are evaluated
        if (LMIN) {
            local_r = s_R[ty][tx];
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) local_r = fminf(local_r, s_R[ty + i][tx + j]);
        }
        racc += local_r;
        if (live && local_r != 0.f) {
#pragma unroll
            for (int sa = 0; sa < 2; ++sa)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const FrameGeo q = frame_geom<GEOM_P2, ISO>(fl, g, pq[sa][sb]);
                    if (q.valid) {
                        const float* __restrict__ rc = s_raw + (q.ci - w.ry0) * RPITCH + (q.cj - w.rx0);
                        const int cx0 = q.x0 - w.cx0, cy0 = q.y0 - w.cy0;
                        const int cx1 = min(q.x0 + 1, g.gw - 1) - w.cx0, cy1 = min(q.y0 + 1, g.gh - 1) - w.cy0;
                        taps_accum<ISO, false>(
                            q, g, local_r, [=](int di, int dj) { return rc[di * RPITCH + dj]; },
                            [=](int k) { return s_cov[(k & 2 ? cy1 : cy0) * CP + (k & 1 ? cx1 : cx0)]; },
                            n4[sa][sb], d4[sa][sb]);
                    }
                }
        }
    }
    if (!live) return;
    if (a.acc_r) a.acc_r[ridx] = ((a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[ridx] : 0.f) + racc;
#pragma unroll
    for (int sa = 0; sa < 2; ++sa)
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const int hi = 2 * ly + sa, hj = 2 * lx + sb;
            if (border_pixel(g, hi, hj)) continue;  // k_merge_border's
            const size_t o = ((size_t)(hi - g.row0) * g.sW + hj) * 3;
            float n3[3] = {0.f, 0.f, 0.f}, d3[3] = {0.f, 0.f, 0.f};
            if (a.flags & HHSR_MERGE_LOAD_ACC) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    n3[k] = num[o + k];
                    d3[k] = den[o + k];
                }
            }
            if (a.flags & HHSR_MERGE_DO_REF) ref_accum_fast<ISO>(a.ref_raw, a.ref_cov, g, hi, hj, n4[sa][sb], d4[sa][sb]);
            classes_to_rgb(cfa, n4[sa][sb], d4[sa][sb], n3, d3);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                num[o + k] = (a.flags & HHSR_MERGE_DIVIDE) ? n3[k] / d3[k] : n3[k];
                if (a.flags & HHSR_MERGE_STORE_DEN) den[o + k] = d3[k];
            }
        }
}



// ---- x2, second generation: one WAVE per Bayer parity class --------------------------------------------------------
// Same tile as k_merge_burst_quad (16 x 16 LR = 32 x 32 HR pixels inside one flow tile, one thread per LR pixel = its
// 2 x 2 HR pixels), but wave w of the workgroup owns the 8 x 8 LR pixels of ONE parity class (row parity w >> 1, column
// parity w & 1).  With the flow shared by the tile, everything that depends on sub-pixel position and parity is then
// wave-uniform per frame — window-centre offsets, tap distances dx / dy, the covariance cell offset and its bilinear
// weights, the CFA class of every tap — and the per-pixel work shrinks to: 3 rows of the raw window (aligned
// ds_read_b64 pairs), 4 covariance cells blended with uniform weights, one inverse, 9 x (2 FMA + min + v_exp_f32 + FMA
// + add), and 8 FMAs into the parity-class accumulators behind a uniform 4-way branch: 216 -> ~110 VALU instructions
// per output pixel and frame.  The reference frame (Alg. 11) runs through the same code as one more "frame" with its own
// uniform geometry (position idx / scale without the half pixel, D7; round-half-even centre; identity fallback of the
// inverse).  Tiles in which ANY window leaves the image (the image perimeter, or flows larger than the distance to it)
// run the general per-pixel body of the first-generation kernel instead — decided once per tile by a lane-parallel
// scan of the frames' flow vectors.
//   * weights are exp2 of the -0.5 log2(e)-scaled quadratic form in ONE v_exp_f32 (no e * e: the border bands, where
//     denormal weights matter, belong to k_merge_border);
//   * the raw window is staged twice, the second copy shifted by one column, so that every sub-pixel of every parity
//     class reads 8-byte aligned pairs (stride-2 dword reads would be 2-way bank conflicts);
//   * the finished 32 x 32 x 3 tile goes through LDS and leaves as whole 16-byte vectors in 384-byte row segments
//     (the per-thread dword stores of the first kernel wrote 1.42 x the output bytes).
// What the A/B builds of rounds 2-4 settled (the losing sides are gone from the source; numbers: DESIGN.md §4 / docs/history):
// clamp(v_exp_f32) + an exact arm for non-finite coefficients instead of min + v_exp_f32 per tap (3.54 vs 3.63 ms); the
// reference frame through run-time selects instead of a compile-time variant of the frame code (160 VGPRs: 3.97 vs 3.54
// ms); per-frame geometry once per workgroup through LDS; 3 + 3 channel accumulators instead of 4 + 4 parity classes;
// k_merge_xs: EDGE frames through the uniform code with masks, the LDS reads of sub-pixel q + 1 issued before the taps of q.
#ifndef HHSR_XS_OCC
#define HHSR_XS_OCC 2  // k_merge_xs<3>: 54 accumulators per thread; 3 waves per SIMD (168 VGPRs) spills 50 dwords
#endif
#ifndef HHSR_X2_OCC
#define HHSR_X2_OCC 4  // waves per SIMD the register allocation of k_merge_x2 is held to (122 VGPRs; A/B: 3 = 4; 5 spills: 6.9 ms)
#endif
constexpr int X2_RP = 24;   // raw / R window pitch in floats: rows are read with stride 2 -> 48 dwords = 16 (mod 32) banks
constexpr int X2_CP = 24;   // covariance window pitch in float4: 96 dwords = 32 (mod 64) banks for ds_read_b128
constexpr int X2_OP = 100;  // output tile pitch in floats (96 + 4: rows stay 16-byte aligned)
constexpr float X2_KEXP = -0.72134752044448170368f;  // -0.5 * log2(e)
constexpr int X2_WIN = QT + 3;                        // 19 x 19 raw window

// LDS reads as exactly the instruction written: the compiler narrows a float4 whose .z is unused into ds_read2_b32
// (cells are 4 dwords apart: 4-way bank conflicts), narrows a half-used pair to a stride-2 ds_read_b32 (2-way) and
// merges neighbouring pairs into ds_read2_b64 (8 LDS cycles instead of 2 x 2) — measured with tools/ubench/lds_patterns:
// 444 LDS cycles per wave and frame instead of ~200, more than half of them bank conflicts.  Volatile keeps the access
// width; the loads are still scheduled and waited for by the compiler (unlike inline asm).
typedef float hhsr_v2f __attribute__((ext_vector_type(2)));
typedef float hhsr_v4f __attribute__((ext_vector_type(4)));
#define HHSR_LDS __attribute__((address_space(3)))
__device__ __forceinline__ float2 lds_pair(const float* p) {  // p: 8-byte aligned LDS address
    const hhsr_v2f v = *(const volatile HHSR_LDS hhsr_v2f*)p;
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ float4 lds_quad(const float4* p) {  // p: LDS address
    const hhsr_v4f v = *(const volatile HHSR_LDS hhsr_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

struct X2Axis {       // wave-uniform geometry of one axis of one frame
    int org;          // raw coordinate of window index 0
    int e[2];         // first tap of sub-pixel s sits at window index t + e[s] (t = the LR pixel's index in the tile)
    float d0[2];      // centre tap minus sampling position (taps_accum's dx0 / dy0)
    int oc[2];        // covariance cell of sub-pixel s = l + oc[s] in the staged cell window (l = lj or li)
    float f[2];       // its bilinear fraction
};

// comp frame (merge.py:319-361): position (h + 0.5)/2 + flow; frame_geom<GEOM_P2> per sub-pixel, bit for bit
__device__ __forceinline__ int x2_comp_org(float fl, int l0) {
    const float fi = floorf(fl);
    return l0 + (int)fi + (int)(fl >= fi + 0.75f) - 1;
}
__device__ __forceinline__ X2Axis x2_comp_axis(float fl, int l0, int p) {
    X2Axis u;
    const float fi = floorf(fl);
    const int c0 = fl >= fi + 0.75f, c1 = fl >= fi + 0.25f;  // exact, see frame_geom<GEOM_P2>
    u.org = l0 + (int)fi + c0 - 1;
    u.e[0] = 0;
    u.e[1] = c1 - c0;
    const float fr0 = (fl - fi) + (0.25f - (float)c0), fr1 = (fl - fi) + (0.75f - (float)c1);
    u.d0[0] = 0.5f - fr0;
    u.d0[1] = 0.5f - fr1;
    // covariance cell x0 = (cj - 1) >> 1, fraction 0.5 ((cj - 1) & 1 + fr), cj - 1 = org + t + e; window origin org >> 1
    const int m0 = (u.org & 1) + p, m1 = m0 + u.e[1];
    u.oc[0] = m0 >> 1;
    u.oc[1] = m1 >> 1;
    u.f[0] = 0.5f * ((float)(m0 & 1) + fr0);
    u.f[1] = 0.5f * ((float)(m1 & 1) + fr1);
    return u;
}
// reference frame (merge.py:113-114, 179-202; ref_accum_fast): position h / 2 = l + s / 2, centre = round-half-even,
// covariance position (pos - 0.5) / 2 with floor + signed fraction; staged with org = l0 - 1, cell origin (l0 - 1) >> 1
__device__ __forceinline__ X2Axis x2_ref_axis(int l0, int p) {
    X2Axis u;
    u.org = l0 - 1;
    u.e[0] = 0;
    u.e[1] = p;                      // l + 0.5 rounds to the even neighbour: l (even l) or l + 1 (odd l)
    u.d0[0] = 0.f;
    u.d0[1] = p ? 0.5f : -0.5f;
    u.oc[0] = p;                     // cell of (l - 0.5) / 2:       l even: l/2 - 1 (f 0.75), l odd: (l-1)/2 (f 0.25)
    u.oc[1] = 1;                     // cell of l / 2:               l even: l/2 (f 0),        l odd: (l-1)/2 (f 0.5)
    u.f[0] = p ? 0.25f : 0.75f;
    u.f[1] = p ? 0.5f : 0.f;
    return u;
}

// the R, G, B sums of one sub-pixel from its NC accumulators (3: channels already; 4: parity classes -> channels)
template <int NC>
__device__ __forceinline__ void xs_rgb(const Cfa4 cfa, const float* nsub, const float* dsub, float n3[3], float d3[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) n3[k] = d3[k] = 0.f;
    if (NC == 3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            n3[k] = nsub[k];
            d3[k] = dsub[k];
        }
    } else {
        const float n4[2][2] = {{nsub[0], nsub[1]}, {nsub[2], nsub[3]}}, d4[2][2] = {{dsub[0], dsub[1]}, {dsub[2], dsub[3]}};
        classes_to_rgb(cfa, n4, d4, n3, d3);
    }
}


static inline bool cfa_is_bayer(const Cfa4& c) {  // red (0) and blue (2) on one diagonal, green (1) on the other
    for (int k = 0; k < 4; ++k)
        if (c.c[k] == 0) return c.c[3 - k] == 2 && c.c[k ^ 1] == 1 && c.c[k ^ 2] == 1;
    return false;
}

// ---- launchers of the kernel families (defined next to their kernels) ---------------------------------------------------
void hhsr_launch_merge_tile(bool p2, bool iso, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g, const Cfa4& c,
                            float* num, float* den);
void hhsr_launch_merge_quad(bool iso, bool lmin, bool mono, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g,
                            const Cfa4& c, float* num, float* den);
void hhsr_launch_merge_x2(bool iso, bool lmin, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g, const Cfa4& c,
                          float* num, float* den);
void hhsr_launch_merge_x3(bool iso, bool lmin, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g, const Cfa4& c,
                          float* num, float* den);

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
                                                         float* __restrict__ den) {
    const int oj = blockIdx.x * 64 + (threadIdx.x & 63), oi = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (oj >= g.sW || oi >= g.sH) return;
    float val[3] = {0.f, 0.f, 0.f}, acc[3] = {0.f, 0.f, 0.f};
    const bool over = ref_contrib<ISO>(raw, cov, g, cfa, oi, oj, acc_rob, rad_max, max_mult, max_fc, val, acc);
    const size_t o = ((size_t)oi * g.sW + oj) * 3;
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

// ---- host entry points ----------------------------------------------------------------------------------
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

template <bool ISO, bool LMIN, bool MONO = false>
__global__ void __launch_bounds__(256) k_merge_burst_quad(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                           float* __restrict__ den) {
    __shared__ float s_raw[RWIN * RPITCH];
    __shared__ float4 s_cov[MONO ? (QT + 3) * CWM : CWIN * CWIN];
    __shared__ float s_R[LMIN ? (QT + 4) * (QT + 4 + 1) : 1];  // LMIN: un-filtered robustness of the tile + 2-pixel border
    quad_tile_body<ISO, LMIN, MONO>(a, g, cfa, num, den, s_raw, s_cov, s_R);
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
#ifndef HHSR_X2_DB
#define HHSR_X2_DB 0   // 1: double-buffered LDS windows, one barrier per frame (A/B measured: see DESIGN.md)
#endif
#ifndef HHSR_X2_CLAMP
#define HHSR_X2_CLAMP 1  // 1: clamp(v_exp_f32) + exact path for non-finite coefficients (A/B: 3.54 ms); 0: min + v_exp_f32 per tap (3.63)
#endif
#ifndef HHSR_X2_PEEL
#define HHSR_X2_PEEL 0   // 1: reference frame as a compile-time variant of the frame code (A/B: 160 VGPRs, 3.97 vs 3.54 ms); 0: run-time selects
#endif
#ifndef HHSR_X2_GEO
#define HHSR_X2_GEO 1  // 1: per-frame geometry evaluated once per workgroup (lane = frame) and broadcast through LDS
#endif
#ifndef HHSR_XS_OCC
#define HHSR_XS_OCC 2  // k_merge_xs<3>: 72 accumulators per thread; 3 waves per SIMD (168 VGPRs) spills 50 dwords
#endif
#ifndef HHSR_X2_RGB
#define HHSR_X2_RGB 1  // k_merge_x2: 3 + 3 channel accumulators per sub-pixel (Bayer) instead of 4 + 4 parity classes
#endif
#ifndef HHSR_XS_EDGE
#define HHSR_XS_EDGE 1  // k_merge_xs: frames whose window leaves the image run the uniform code with masks (0: per-pixel path)
#endif
#ifndef HHSR_XS_RGB
#define HHSR_XS_RGB 1  // k_merge_xs: 3 + 3 channel accumulators per sub-pixel (Bayer) instead of 4 + 4 parity classes
#endif
#ifndef HHSR_XS_PIPE
#define HHSR_XS_PIPE 1  // k_merge_xs: LDS reads of sub-pixel q + 1 issued before the taps of sub-pixel q
#endif
#ifndef HHSR_X2_OCC
#define HHSR_X2_OCC 4  // waves per SIMD the register allocation of k_merge_x2 is held to (125 VGPRs; A/B: 3 = 4; 5 spills: 6.9 ms)
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

template <bool ISO, bool LMIN>
__global__ void __launch_bounds__(256, HHSR_X2_OCC) k_merge_x2(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                   float* __restrict__ den) {
    constexpr int NB = HHSR_X2_DB ? 2 : 1;                           // window buffers (2: one barrier per frame)
    constexpr int RAWSZ = 20 * X2_RP, COVSZ = CWIN * X2_CP;
    __shared__ __align__(16) float s_rawA[NB * RAWSZ];              // window[y][x]
    __shared__ __align__(16) float s_rawB[NB * RAWSZ];              // window[y][x + 1]
    __shared__ float4 s_cov[NB * COVSZ];
    __shared__ __align__(16) float s_R[NB * RAWSZ];                 // LMIN: un-filtered robustness, tile + 2-pixel border
    __shared__ __align__(16) float s_out[32 * X2_OP];
#if HHSR_X2_GEO
    __shared__ float4 s_geo[(HHSR_MAX_FRAMES + 1) * 8];             // per frame: [axis x, y][parity 0, 1] x 2 quads
#endif
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // readfirstlane: known wave-uniform
    const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
    const int bid = xcd_remap(blockIdx.y * nbx + blockIdx.x, nblk);
    const int lx0 = (bid % nbx) * QT, ly0 = (g.row0 >> 1) + (bid / nbx) * QT;  // LR origin of the workgroup
    const int lrow1 = g.row1 >> 1;
    const int tile = (ly0 / g.ts) * g.nx + lx0 / g.ts;

    // ---- can the whole tile take the uniform path?  every window of every frame inside the image -------------------
    // Chained launches (bursts whose last frames arrive late: graph.HostBurstRunner).  A STORE_CLASSES launch merges the
    // frames it has into the parity-class accumulators and parks them in a.cls; a LOAD_CLASSES launch gets all frames so
    // far, restores the accumulators and continues with frames [a.first, a.n) — a middle link parks them again, the final
    // link adds the reference frame and runs the epilogue.  The register contents carry over exactly, so the result is
    // bit-identical to ONE launch over all frames — provided a tile runs the same code in every link: the storing links
    // apply the reference frame's border rule too and simply skip the tiles they would send down the per-pixel path
    // (once a frame's window leaves the image the tile stays skipped: the set of frames only grows); the final link decides
    // over ALL frames (like the single launch) and recomputes its per-pixel tiles from the first frame on (2 % of the
    // tiles at 12 MP).
    const bool chain_store = (a.flags & HHSR_MERGE_STORE_CLASSES) != 0, chain_load = (a.flags & HHSR_MERGE_LOAD_CLASSES) != 0;
    bool ok = lx0 + QT <= g.W && ly0 + QT <= lrow1;
    if (((a.flags & HHSR_MERGE_DO_REF) || chain_store) &&
        !(lx0 >= 1 && lx0 + QT + 2 <= g.W && ly0 >= 1 && ly0 + QT + 2 <= g.H)) ok = false;
    if (ok && lane < a.n) {
        const float2 fl = a.f[lane].flow[tile];
        const int ox = x2_comp_org(fl.x, lx0), oy = x2_comp_org(fl.y, ly0);
        ok = ox >= 0 && ox + X2_WIN <= g.W && oy >= 0 && oy + X2_WIN <= g.H;  // NaN flow: (int) of NaN is checked too
        ok = ok && fl.x == fl.x && fl.y == fl.y;
    }
    if (!__all(ok)) {  // wave-uniform, identical in the four waves
        if (chain_store) return;  // (the final link recomputes this tile over all frames)
        quad_tile_body<ISO, LMIN>(a, g, cfa, num, den, s_rawA, s_cov, s_R);
        return;
    }
    float* __restrict__ cls = a.cls ? a.cls + (size_t)bid * (33 * 256) + tid : nullptr;
    const int nfirst = chain_load ? a.first : 0;

#if HHSR_X2_GEO
    // Per-frame geometry once per WORKGROUP: it only depends on the frame's flow vector and the parity class, so
    // evaluating it in every thread and frame (~50 instructions, ~40 % of them half-rate, identical in all lanes of a
    // wave) was 7 % of the kernel's VALU time.  Lane = frame here; the frame loop reads its entry back with four
    // broadcast ds_read_b128.  (Visible to everybody after the first barrier of the frame loop.)
    for (int n = tid; n < a.n + ((a.flags & HHSR_MERGE_DO_REF) ? 1 : 0); n += 256) {
        const bool isref = n >= a.n;
        float2 fl = make_float2(0.f, 0.f);
        if (!isref) fl = a.f[n].flow[tile];
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {
            const float f = axis ? fl.y : fl.x;
            const int l0 = axis ? ly0 : lx0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const X2Axis u = isref ? x2_ref_axis(l0, p) : x2_comp_axis(f, l0, p);
                s_geo[n * 8 + axis * 4 + p * 2] = make_float4(__int_as_float(u.org), __int_as_float(u.e[1]), u.d0[0], u.d0[1]);
                s_geo[n * 8 + axis * 4 + p * 2 + 1] = make_float4(__int_as_float(u.oc[0]), __int_as_float(u.oc[1]), u.f[0], u.f[1]);
            }
        }
    }
#endif
    const int py = wave >> 1, px = wave & 1;                        // this wave's parity class
    const int li = lane >> 3, lj = lane & 7;
    const int ty = 2 * li + py, tx = 2 * lj + px;                   // LR pixel inside the tile
    const int ridx = (ly0 + ty) * g.W + lx0 + tx;
    // HHSR_X2_RGB (round 4; Bayer layouts — the only ones this kernel is launched for): the two green parity classes are
    // summed when a frame is folded: 3 + 3 accumulators per sub-pixel instead of 4 + 4 (24 instead of 32 per thread),
    // and the epilogue has no class -> channel step (whose private arrays lived in scratch).  0: round 3's four classes.
    constexpr int NC = HHSR_X2_RGB ? 3 : 4, NA = 4 * NC;
    float nacc[2][2][NC], dacc[2][2][NC];
    const int rcl = cfa.c[0] == 0 ? 0 : cfa.c[1] == 0 ? 1 : cfa.c[2] == 0 ? 2 : 3;  // parity class of the red samples
    const int ri = rcl >> 1, rj = rcl & 1;
    float racc = 0.f;
    if (chain_load) {  // (coalesced: 256 consecutive floats per accumulator and tile)
#pragma unroll
        for (int k = 0; k < NA; ++k) {  // (the parking layout keeps 16 + 16 + 1 slots per thread)
            (&nacc[0][0][0])[k] = cls[k * 256];
            (&dacc[0][0][0])[k] = cls[(16 + k) * 256];
        }
        racc = cls[32 * 256];
    } else {
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            (&nacc[0][0][0])[k] = 0.f;
            (&dacc[0][0][0])[k] = 0.f;
        }
    }

    // staging slots (by thread id, independent of the pixel mapping)
    constexpr int rwin = X2_WIN, cwin = QT / 2 + 3;  // 19 raw pixels, 11 covariance cells
    const int e0 = tid, e1 = tid + 256;
    const int e0y = e0 / rwin, e0x = e0 - e0y * rwin, e1y = e1 / rwin, e1x = e1 - e1y * rwin;
    const int cey = tid / cwin, cex = tid - cey * cwin;
    const bool has1 = e1 < rwin * rwin, hasc = tid < cwin * cwin;
    constexpr int RW = QT + 4;
    const int m0y = tid / RW, m0x = tid - m0y * RW;
    const int m1 = tid + 256, m1y = m1 / RW, m1x = m1 - m1y * RW;
    const bool hasm1 = LMIN && m1 < RW * RW;
    const int moff0 = clampi(ly0 - 2 + m0y, 0, g.H - 1) * g.W + clampi(lx0 - 2 + m0x, 0, g.W - 1);
    const int moff1 = clampi(ly0 - 2 + m1y, 0, g.H - 1) * g.W + clampi(lx0 - 2 + m1x, 0, g.W - 1);
    const int nloop = a.n + ((a.flags & HHSR_MERGE_DO_REF) ? 1 : 0);  // the reference frame is the last "frame"
    float pr0 = 0.f, pr1 = 0.f, plr = 0.f, plr1 = 0.f;
    float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 pfl = make_float2(0.f, 0.f);
    auto prefetch = [&](int n) {  // all windows are inside the image (checked above): no bounds tests
        const bool isref = n >= a.n;
        const float* __restrict__ raw = isref ? a.ref_raw : a.f[n].raw;
        const float4* __restrict__ cov = isref ? a.ref_cov : a.f[n].cov;
        int ox = lx0 - 1, oy = ly0 - 1;
        if (!isref) {
            pfl = a.f[n].flow[tile];
            ox = x2_comp_org(pfl.x, lx0);
            oy = x2_comp_org(pfl.y, ly0);
        }
        pr0 = raw[(size_t)(oy + e0y) * g.pitch + ox + e0x];
        if (has1) pr1 = raw[(size_t)(oy + e1y) * g.pitch + ox + e1x];
        if (!ISO && hasc) pc = cov[(size_t)min((oy >> 1) + cey, g.gh - 1) * g.gw + min((ox >> 1) + cex, g.gw - 1)];
        if (!isref) {
            if (LMIN) {
                plr = a.f[n].r[moff0];
                if (hasm1) plr1 = a.f[n].r[moff1];
            } else {
                plr = a.f[n].r[ridx];
            }
        }
    };

    const float* __restrict__ rbase = s_R + ty * X2_RP + 2 * lj;
    const int cbase = li * X2_CP + lj;

    // write the prefetched registers of one frame into window buffer `bo`; returns that frame's flow / robustness
    float2 sfl = make_float2(0.f, 0.f);
    float sr = 0.f;
    auto stage = [&](int n, int bo) {
        const bool isref = n >= a.n;
        s_rawA[bo * RAWSZ + e0y * X2_RP + e0x] = pr0;
        if (e0x > 0) s_rawB[bo * RAWSZ + e0y * X2_RP + e0x - 1] = pr0;
        if (has1) {
            s_rawA[bo * RAWSZ + e1y * X2_RP + e1x] = pr1;
            if (e1x > 0) s_rawB[bo * RAWSZ + e1y * X2_RP + e1x - 1] = pr1;
        }
        if (!ISO && hasc) s_cov[bo * COVSZ + cey * X2_CP + cex] = pc;
        if (LMIN && !isref) {
            s_R[bo * RAWSZ + m0y * X2_RP + m0x] = plr;
            if (hasm1) s_R[bo * RAWSZ + m1y * X2_RP + m1x] = plr1;
        }
        sfl = pfl;
        sr = isref ? 1.f : plr;
    };
    // One frame of taps.  ISREF is a compile-time flag (the reference frame runs the same code with its own uniform
    // geometry, the identity fallback of the inverse and r = 1): as a run-time select it costs ~40 v_cndmask per frame,
    // and on gfx950 v_cndmask / v_min / v_cmp / v_floor / v_cvt issue at HALF the v_fma rate, v_exp / v_rcp at a quarter
    // (tools/ubench/valu_rate.hip) — the kernel is VALU-bound, so instruction classes are what to count.
    auto frame = [&](auto isref_c, const bool isref_rt, const float2 fl, float local_r, const int bo, const int n) {
        const bool isref = HHSR_X2_PEEL ? decltype(isref_c)::value : isref_rt;
        if (LMIN && !isref) {  // 5 x 5 minimum over rows ty .. ty + 4, columns tx .. tx + 4 of the R window
            // R is clamped to [0, 1] (never negative, never NaN): the order of its float32 bit patterns is the order of
            // the values, and v_min3_u32 needs no canonicalisation of its inputs (fminf costs a v_max per operand: 28
            // half-rate instructions per thread and frame here)
            unsigned m = 0x7f7fffffu;
            if (px) {  // (uniform branch instead of a select per row)
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const float2 v01 = lds_pair(rbase + bo * RAWSZ + r * X2_RP), v23 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 2);
                    const float2 v45 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 4);
                    m = min(m, min(min(__float_as_uint(v01.y), __float_as_uint(v23.x)),
                                   min(__float_as_uint(v23.y), min(__float_as_uint(v45.x), __float_as_uint(v45.y)))));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const float2 v01 = lds_pair(rbase + bo * RAWSZ + r * X2_RP), v23 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 2);
                    const float2 v45 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 4);
                    m = min(m, min(min(__float_as_uint(v01.y), __float_as_uint(v23.x)),
                                   min(__float_as_uint(v23.y), min(__float_as_uint(v45.x), __float_as_uint(v01.x)))));
                }
            }
            local_r = __uint_as_float(m);
        }
        if (!isref) racc += local_r;
        if (local_r == 0.f) return;
#if HHSR_X2_GEO
        X2Axis ax, ay;
        {
            const float4 xa = lds_quad(s_geo + n * 8 + px * 2), xb = lds_quad(s_geo + n * 8 + px * 2 + 1);
            const float4 ya = lds_quad(s_geo + n * 8 + 4 + py * 2), yb = lds_quad(s_geo + n * 8 + 4 + py * 2 + 1);
            ax.org = __float_as_int(xa.x); ax.e[0] = 0; ax.e[1] = __float_as_int(xa.y); ax.d0[0] = xa.z; ax.d0[1] = xa.w;
            ax.oc[0] = __float_as_int(xb.x); ax.oc[1] = __float_as_int(xb.y); ax.f[0] = xb.z; ax.f[1] = xb.w;
            ay.org = __float_as_int(ya.x); ay.e[0] = 0; ay.e[1] = __float_as_int(ya.y); ay.d0[0] = ya.z; ay.d0[1] = ya.w;
            ay.oc[0] = __float_as_int(yb.x); ay.oc[1] = __float_as_int(yb.y); ay.f[0] = yb.z; ay.f[1] = yb.w;
        }
#else
        const X2Axis ax = isref ? x2_ref_axis(lx0, px) : x2_comp_axis(fl.x, lx0, px);
        const X2Axis ay = isref ? x2_ref_axis(ly0, py) : x2_comp_axis(fl.y, ly0, py);
#endif
#pragma unroll
        for (int sa = 0; sa < 2; ++sa)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                float ixx = 2.f * X2_KEXP, ixy = 0.f, iyy = 2.f * X2_KEXP;
                bool finite = true;
                if (!ISO) {
                    const int ca = cbase + ay.oc[sa] * X2_CP + ax.oc[sb];
                    const float4 c00 = lds_quad(s_cov + bo * COVSZ + ca), c01 = lds_quad(s_cov + bo * COVSZ + ca + 1);
                    const float4 c10 = lds_quad(s_cov + bo * COVSZ + ca + X2_CP), c11 = lds_quad(s_cov + bo * COVSZ + ca + X2_CP + 1);
                    const float gx = ax.f[sb], gy = ay.f[sa];
                    const float w11 = gx * gy, w01 = gx - w11, w10 = gy - w11, w00 = (1.f - gx) - w10;
                    const float cxx = fmaf(w11, c11.x, fmaf(w10, c10.x, fmaf(w01, c01.x, w00 * c00.x)));
                    const float cxy = fmaf(w11, c11.y, fmaf(w10, c10.y, fmaf(w01, c01.y, w00 * c00.y)));
                    const float cyy = fmaf(w11, c11.w, fmaf(w10, c10.w, fmaf(w01, c01.w, w00 * c00.w)));
                    const float det = fmaf(cxx, cyy, -(cxy * cxy));
                    const float s1 = __builtin_amdgcn_rcpf(det) * X2_KEXP;
                    ixx = s1 * cyy;
                    ixy = (-2.f * s1) * cxy;
                    iyy = s1 * cxx;
                    if (isref && !(fabsf(det) > 1e-10f)) {  // linalg.py:53-64: identity (also for NaN, D10)
                        ixx = X2_KEXP;
                        ixy = 0.f;
                        iyy = X2_KEXP;
                    }
                    // 0 * x is 0 for finite x and NaN for NaN / inf: one NaN test for the three coefficients
                    if (HHSR_X2_CLAMP) {
                        const float probe = fmaf(0.f, ixx, fmaf(0.f, ixy, 0.f * iyy));
                        finite = probe == probe;
                    }
                }
                // the 3 x 3 taps: rows ty + e .. + 2, columns tx + e .. + 2 of the window, as aligned pairs from the
                // copy whose shift makes column tx + e even
                const int mcol = px + ax.e[sb];  // 0, 1, 2
                const float* __restrict__ rp = ((mcol & 1) ? s_rawB : s_rawA) + bo * RAWSZ + (ty + ay.e[sa]) * X2_RP + 2 * lj + (mcol & 2);
                const float dx0 = ax.d0[sb], dy0 = ay.d0[sa];
                const float dxs[3] = {dx0 - 1.f, dx0, dx0 + 1.f};
                float sv[2][2], sd[2][2];  // by parity of the tap offset (di + 1, dj + 1)
                // w = exp(-max(0, q) / 2) with Python's max (NaN -> 0, D10).  With finite coefficients q is finite and
                // w = clamp(exp2(z), 0, 1): the clamp is an output modifier of v_exp_f32 (free) and equals the max for
                // z > 0 (non-positive-definite blends at the image border, D11).  Non-finite coefficients (NaN
                // covariances of flat regions, D10; singular hand-made covariances) take the exact per-tap form.
                auto taps = [&](auto exact_c) {
                    constexpr bool EXACT = decltype(exact_c)::value;
#pragma unroll
                    for (int di = 0; di < 3; ++di) {
                        const float2 v01 = lds_pair(rp + di * X2_RP), v23 = lds_pair(rp + di * X2_RP + 2);
                        const float c3[3] = {v01.x, v01.y, v23.x};
                        const float dy = dy0 + (float)(di - 1);
                        const float qa = iyy * dy * dy, qb = ixy * dy;
#pragma unroll
                        for (int dj = 0; dj < 3; ++dj) {
                            const float dx = dxs[dj];
                            const float z = fmaf(fmaf(ixx, dx, qb), dx, qa);
                            const float w = EXACT ? __builtin_amdgcn_exp2f(fminf(z, 0.f))
                                                  : __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(z), 0.f, 1.f);
                            if (di < 2 && dj < 2) {  // first tap of its parity class (row-major order)
                                sv[di & 1][dj & 1] = w * c3[dj];
                                sd[di & 1][dj & 1] = w;
                            } else {
                                sv[di & 1][dj & 1] = fmaf(w, c3[dj], sv[di & 1][dj & 1]);
                                sd[di & 1][dj & 1] += w;
                            }
                        }
                    }
                };
                if (!HHSR_X2_CLAMP) taps(std::true_type{});
                else if (ISO || finite) taps(std::false_type{});
                else taps(std::true_type{});
                // tap-offset parity -> absolute raw-coordinate parity (uniform): class (a, b) += r * sv[a ^ by][b ^ bx]
                const int by = (ay.org + py + ay.e[sa]) & 1, bx = (ax.org + px + ax.e[sb]) & 1;
                // (the empty asm statements keep the four arms real branches: if-converted, the permutation costs 16
                // v_cndmask per sub-pixel, twice the FMAs it feeds;
                // and distinct, so that the FMAs are not sunk below the arms leaving 8 permutation moves in each)
#if HHSR_X2_RGB
                // tap parity (a, b) is colour class (a ^ by, b ^ bx): red sits at parity (ri ^ by, rj ^ bx), blue diagonally
                // opposite, the greens on the other diagonal — four wave-uniform arrangements
                const int ra = ri ^ by, rb = rj ^ bx;
#define HHSR_FOLD3(RA, RB)                                                                    \
    {                                                                                         \
        nacc[sa][sb][0] = fmaf(local_r, sv[RA][RB], nacc[sa][sb][0]);                         \
        dacc[sa][sb][0] = fmaf(local_r, sd[RA][RB], dacc[sa][sb][0]);                         \
        nacc[sa][sb][1] = fmaf(local_r, sv[RA ^ 1][RB] + sv[RA][RB ^ 1], nacc[sa][sb][1]);    \
        dacc[sa][sb][1] = fmaf(local_r, sd[RA ^ 1][RB] + sd[RA][RB ^ 1], dacc[sa][sb][1]);    \
        nacc[sa][sb][2] = fmaf(local_r, sv[RA ^ 1][RB ^ 1], nacc[sa][sb][2]);                 \
        dacc[sa][sb][2] = fmaf(local_r, sd[RA ^ 1][RB ^ 1], dacc[sa][sb][2]);                 \
    }
                if (ra) {
                    if (rb) { asm volatile("; fold 11"); HHSR_FOLD3(1, 1) asm volatile("; end 11"); }
                    else { asm volatile("; fold 10"); HHSR_FOLD3(1, 0) asm volatile("; end 10"); }
                } else {
                    if (rb) { asm volatile("; fold 01"); HHSR_FOLD3(0, 1) asm volatile("; end 01"); }
                    else { asm volatile("; fold 00"); HHSR_FOLD3(0, 0) asm volatile("; end 00"); }
                }
#undef HHSR_FOLD3
#else
#define HHSR_FOLD(BY, BX)                                                                             \
    _Pragma("unroll") for (int aa = 0; aa < 2; ++aa) _Pragma("unroll") for (int bb = 0; bb < 2; ++bb) { \
        nacc[sa][sb][aa * 2 + bb] = fmaf(local_r, sv[aa ^ BY][bb ^ BX], nacc[sa][sb][aa * 2 + bb]);    \
        dacc[sa][sb][aa * 2 + bb] = fmaf(local_r, sd[aa ^ BY][bb ^ BX], dacc[sa][sb][aa * 2 + bb]);    \
    }
                if (by) {
                    if (bx) { asm volatile("; fold 11"); HHSR_FOLD(1, 1) asm volatile("; end 11"); }
                    else { asm volatile("; fold 10"); HHSR_FOLD(1, 0) asm volatile("; end 10"); }
                } else {
                    if (bx) { asm volatile("; fold 01"); HHSR_FOLD(0, 1) asm volatile("; end 01"); }
                    else { asm volatile("; fold 00"); HHSR_FOLD(0, 0) asm volatile("; end 00"); }
                }
#undef HHSR_FOLD
#endif
            }
    };
    auto frame_n = [&](int n, const float2 fl, float lr, int bo) {
        if (!HHSR_X2_PEEL) frame(std::false_type{}, n >= a.n, fl, lr, bo, n);
        else if (n >= a.n) frame(std::true_type{}, true, fl, lr, bo, n);
        else frame(std::false_type{}, false, fl, lr, bo, n);
    };
#if HHSR_X2_DB
    // double-buffered windows: frame n is evaluated from buffer n & 1 while frame n + 1 is written into the other one
    // (its global loads were issued before frame n's taps): ONE workgroup barrier per frame
    if (nloop > 0) {
        prefetch(0);
        stage(0, 0);
        __syncthreads();
        if (nloop > 1) prefetch(1);
    }
    for (int n = 0; n < nloop; ++n) {
        const float2 fl = sfl;
        const float lr = sr;
        frame_n(n, fl, lr, n & 1);
        if (n + 1 < nloop) stage(n + 1, (n + 1) & 1);
        __syncthreads();
        if (n + 2 < nloop) prefetch(n + 2);
    }
#else
    if (nloop > nfirst) prefetch(nfirst);
    for (int n = nfirst; n < nloop; ++n) {
        __syncthreads();  // the previous frame's taps are done with the LDS windows
        stage(n, 0);
        const float2 fl = sfl;
        const float lr = sr;
        __syncthreads();
        if (n + 1 < nloop) prefetch(n + 1);  // in flight while this frame's taps are evaluated
        frame_n(n, fl, lr, 0);
    }
#endif
    if (chain_store) {  // park the accumulators for the final link
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            cls[k * 256] = (&nacc[0][0][0])[k];
            cls[(16 + k) * 256] = (&dacc[0][0][0])[k];
        }
        cls[32 * 256] = racc;
        return;
    }
    if (a.acc_r) a.acc_r[ridx] = ((a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[ridx] : 0.f) + racc;
    // ---- epilogue: CFA classes -> RGB, normalise, store -----------------------------------------------------------------
    const int ly = ly0 + ty, lx = lx0 + tx;
    if (a.flags & HHSR_MERGE_LOAD_ACC) {
        // chained launches (bursts longer than one launch, multi-GPU finish): per-pixel read-modify-write; the border
        // bands keep their input for k_merge_border
#pragma unroll
        for (int sa = 0; sa < 2; ++sa)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int hi = 2 * ly + sa, hj = 2 * lx + sb;
                if (border_pixel(g, hi, hj)) continue;
                const size_t o = ((size_t)(hi - g.row0) * g.sW + hj) * 3;
                float n3[3], d3[3];
                xs_rgb<NC>(cfa, nacc[sa][sb], dacc[sa][sb], n3, d3);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float nk = num[o + k] + n3[k], dk = den[o + k] + d3[k];
                    num[o + k] = (a.flags & HHSR_MERGE_DIVIDE) ? nk / dk : nk;
                    if (a.flags & HHSR_MERGE_STORE_DEN) den[o + k] = dk;
                }
            }
        return;
    }
    // whole tile through LDS: rows of 96 floats leave as float4 (k_merge_border overwrites the border bands afterwards)
    const int npass = (a.flags & HHSR_MERGE_STORE_DEN) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int sa = 0; sa < 2; ++sa) {
            float v[2][3];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                float n3[3], d3[3];
                xs_rgb<NC>(cfa, nacc[sa][sb], dacc[sa][sb], n3, d3);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    v[sb][k] = pass ? d3[k] : ((a.flags & HHSR_MERGE_DIVIDE) ? n3[k] / d3[k] : n3[k]);
            }
            float* row = s_out + (2 * ty + sa) * X2_OP + 6 * tx;
            *reinterpret_cast<float2*>(row) = make_float2(v[0][0], v[0][1]);
            *reinterpret_cast<float2*>(row + 2) = make_float2(v[0][2], v[1][0]);
            *reinterpret_cast<float2*>(row + 4) = make_float2(v[1][1], v[1][2]);
        }
        __syncthreads();
        float* __restrict__ dst = pass ? den : num;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int qd = tid + 256 * r;
            const int orow = qd / 24, oc = (qd - orow * 24) * 4;
            *reinterpret_cast<float4*>(dst + ((size_t)(2 * ly0 + orow - g.row0) * g.sW + 2 * lx0) * 3 + oc) =
                *reinterpret_cast<const float4*>(s_out + orow * X2_OP + oc);
        }
    }
}

// ---- integer scales S >= 2 in general: the wave-per-parity-class kernel with S x S sub-pixels per thread ----------------
// k_merge_x2's design does not need a power-of-two scale: the per-frame geometry is wave-uniform, so it can afford the
// reference's float64 evaluation of (h + 0.5)/S + flow (merge.py:319-345) — a handful of float64 operations per thread
// and frame instead of ~25 per output pixel and frame in k_merge_burst_tile<GEOM_F64>, whose ~250 VALU instructions per
// pixel-frame made x3 at 48 MP a 45 ms merge (C5).  Window centres: h = S l + s, lr = l + (2 s + 1)/(2 S); the centre
// advances by one raw pixel when frac(flow) >= t_s = (2 S - 2 s - 1)/(2 S), decided in float64 where
// frac(flow) = flow - floor(flow) is exact (it is NOT exact in float32 for small negative flows).  t_s is either exactly
// 1/2 or not a binary fraction; a float32 frac(flow) is then never closer to it than ~5e-9 while the reference's float64
// evaluation of lr + flow is off by < 1e-13: same decision, for every l.  The reference frame keeps its position
// h / S in FLOAT32 (merge.py:113-114), which is not the same for all l — its tap distances and covariance fractions are
// therefore per-thread values here (only that one "frame" pays for it).
// Tile = 16 x 16 LR = 16 S x 16 S HR pixels inside one flow tile (ts % 16 == 0); S^2 x 8 accumulators per thread.
// (k_merge_x2's per-workgroup geometry table was tried here too: no change, 41.2 ms either way — at 2 waves per SIMD
// this kernel waits on LDS latency, not on VALU issue.)
// (Measured alternative, round 2: one workgroup per tile AND output sub-row — 24 accumulators per thread, 3 waves per
// SIMD without spills — is slower, 45.2 ms against 41.1 ms at C5: the per-frame work that does not depend on the
// sub-row (staging, 5 x 5 minimum, wave-uniform float64 geometry) is then paid three times and outweighs the occupancy;
// at 4 waves per SIMD it spills 46 dwords: 85 ms.)
// Tiles with a window outside the image run the generic per-pixel code (merge_pixel) from global memory.
template <int S>
struct XsAxis {
    int org;        // raw coordinate of window index 0
    int e[S];       // first tap of sub-pixel s at window index t + e[s]
    float d0[S];    // centre tap minus sampling position
    int oc[S];      // covariance cell of sub-pixel s = l + oc[s] in the staged cell window
    float f[S];     // its bilinear fraction
};

template <int S>
__device__ __forceinline__ int xs_comp_org(float fl, int l0) {
    const float fi = floorf(fl);
    const double frac = (double)fl - (double)fi;  // exact
    return l0 + (int)fi + (int)(frac >= (double)(2 * S - 1) / (double)(2 * S)) - 1;
}

template <int S>
__device__ __forceinline__ XsAxis<S> xs_comp_axis(float fl, int l0, int p) {
    XsAxis<S> u;
    const float fi = floorf(fl);
    const double frac = (double)fl - (double)fi;  // exact
    int c[S];
    float fr[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        c[s] = frac >= (double)(2 * S - 2 * s - 1) / (double)(2 * S);
        fr[s] = (float)(frac + ((double)(2 * s + 1) / (double)(2 * S) - (double)c[s]));  // lr + flow - centre, in [0, 1)
    }
    u.org = l0 + (int)fi + c[0] - 1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        u.e[s] = c[s] - c[0];
        u.d0[s] = 0.5f - fr[s];
        const int m = (u.org & 1) + p + u.e[s];  // cj - 1 = org + t + e, t = 2 lj + p; cell window origin org >> 1
        u.oc[s] = m >> 1;
        u.f[s] = 0.5f * ((float)(m & 1) + fr[s]);
    }
    return u;
}

// reference frame, per thread: l = the thread's LR coordinate, t = l - l0 (ref_accum_fast's arithmetic)
template <int S>
__device__ __forceinline__ XsAxis<S> xs_ref_axis(int l, int l0, int off_lr, double scale, int lcell) {
    XsAxis<S> u;
    u.org = l0 - 1;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float pos = (float)((double)(S * (l + off_lr) + s) / scale) - (float)off_lr;
        const int c = (int)rintf(pos);
        u.e[s] = c - l;  // tap start c - 1 = org + (l - l0) + e
        u.d0[s] = (float)c - pos;
        const float gq = (pos - 0.5f) * 0.5f;
        u.oc[s] = (int)fmaxf(floorf(gq), 0.f) - (u.org >> 1) - lcell;
        u.f[s] = gq - truncf(gq);
    }
    return u;
}

template <int S, bool ISO, bool LMIN>
__global__ void __launch_bounds__(256, HHSR_XS_OCC) k_merge_xs(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                      float* __restrict__ den) {
    constexpr int RAWSZ = 20 * X2_RP, COVSZ = CWIN * X2_CP, OP = 3 * S * QT + 4, OROWS = S * QT;
    __shared__ __align__(16) float s_rawA[RAWSZ];
    __shared__ __align__(16) float s_rawB[RAWSZ];
    __shared__ float4 s_cov[COVSZ];
    __shared__ __align__(16) float s_R[RAWSZ];
    __shared__ __align__(16) float s_out[OROWS * OP];
    __shared__ __align__(16) float s_mskA[RAWSZ];  // EDGE frames: 1 where the window position lies inside the frame, else 0
    __shared__ __align__(16) float s_mskB[RAWSZ];  // (the same shifted by one column, like s_rawB)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
    const int bid = xcd_remap(blockIdx.y * nbx + blockIdx.x, nblk);
    const int lx0 = (bid % nbx) * QT, ly0 = g.row0 / S + (bid / nbx) * QT;  // LR origin of the workgroup
    const int lrow1 = g.row1 / S;
    const int tile = (min(ly0, g.H - 1) / g.ts) * g.nx + min(lx0, g.W - 1) / g.ts;
    const int py = wave >> 1, px = wave & 1;
    const int li = lane >> 3, lj = lane & 7;
    const int ty = 2 * li + py, tx = 2 * lj + px;
    const int ly = ly0 + ty, lx = lx0 + tx;

    // EDGE frames (round 4): a frame whose 19 x 19 window leaves the image — the image's perimeter tiles for the
    // reference frame, tiles pushed over the border by their flow for the others — used to send the whole tile down the
    // per-pixel path below (operands from global memory, float64 geometry per tap: ~40 x the time of a uniform tile;
    // measured at 48 MP x 20: the top and bottom tile rows alone were 5.4 of the launch's 41.4 ms, the perimeter ~9 ms).
    // Now such a frame is staged with clamped coordinates plus a 0 / 1 mask of the window positions inside the frame and
    // evaluated by the SAME uniform code with the reference's border rules applied per lane: taps outside the frame get
    // weight 0 (merge.py:404-405), a sub-pixel whose position lies outside the frame contributes nothing (:346-347), and
    // a centre in column / row 0 takes covariance cells 0 and 1 with the negative fraction (D11, :349-361).  Only partial
    // tiles and non-finite / absurd flows are left to the per-pixel path.
    bool ok = lx0 + QT <= g.W && ly0 + QT <= lrow1;
    bool edge_f = false;
    if (ok && lane < a.n) {
        const float2 fl = a.f[lane].flow[tile];
        ok = fabsf(fl.x) < 1.0e6f && fabsf(fl.y) < 1.0e6f;  // (NaN fails)
        if (ok) {
            const int ox = xs_comp_org<S>(fl.x, lx0), oy = xs_comp_org<S>(fl.y, ly0);
            edge_f = !(ox >= 0 && ox + X2_WIN <= g.W && oy >= 0 && oy + X2_WIN <= g.H);
        }
    }
#if !HHSR_XS_EDGE  // A/B: round 3's rule — any window outside the image sends the tile down the per-pixel path
    if (edge_f || ((a.flags & HHSR_MERGE_DO_REF) && !(lx0 >= 1 && lx0 + QT + 2 <= g.W && ly0 >= 1 && ly0 + QT + 2 <= g.H))) ok = false;
    const unsigned long long edge_mask = 0ull;
    const bool edge_ref = false;
#else
    const unsigned long long edge_mask = __ballot(edge_f);  // bit n: frame n is an EDGE frame (identical in the four waves)
    const bool edge_ref = !(lx0 >= 1 && lx0 + QT + 2 <= g.W && ly0 >= 1 && ly0 + QT + 2 <= g.H);
#endif
    if (!__all(ok)) {
        // generic per-pixel code from global memory for the S x S output pixels of this thread's LR pixel
        if (lx >= g.W || ly >= lrow1) return;
        if (a.acc_r) {
            float racc = (a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[(size_t)ly * g.W + lx] : 0.f;
            for (int n = 0; n < a.n; ++n) racc += robustness_at(a.f[n].r, g, ly, lx, LMIN);
            a.acc_r[(size_t)ly * g.W + lx] = racc;
        }
#pragma unroll 1
        for (int q = 0; q < S * S; ++q) {
            const int hi = S * ly + q / S, hj = S * lx + q % S;
            if (!border_pixel(g, hi, hj)) merge_pixel<float, GEOM_F64, ISO>(a, g, cfa, hi, hj, num, den);
        }
        return;
    }

    const int ridx = ly * g.W + lx;
    // HHSR_XS_RGB (Bayer sensors: the only layouts this kernel is launched for, cfa_is_bayer): the two green parity
    // classes are summed when a frame is folded, so a sub-pixel has 3 + 3 accumulators instead of 4 + 4 (54 instead of 72
    // per thread); 0: the four parity classes of round 3, mapped to channels in the epilogue (A/B)
    constexpr int NC = HHSR_XS_RGB ? 3 : 4;
    float nacc[S][S][NC], dacc[S][S][NC];
#pragma unroll
    for (int k = 0; k < S * S * NC; ++k) {
        (&nacc[0][0][0])[k] = 0.f;
        (&dacc[0][0][0])[k] = 0.f;
    }
    float racc = 0.f;
    // class index of the red (channel 0) sample; blue is the other end of that diagonal, the two greens the other diagonal
    const int rcl = cfa.c[0] == 0 ? 0 : cfa.c[1] == 0 ? 1 : cfa.c[2] == 0 ? 2 : 3;
    const int ri = rcl >> 1, rj = rcl & 1;

    constexpr int rwin = X2_WIN, cwin = QT / 2 + 3;
    const int e0 = tid, e1 = tid + 256;
    const int e0y = e0 / rwin, e0x = e0 - e0y * rwin, e1y = e1 / rwin, e1x = e1 - e1y * rwin;
    const int cey = tid / cwin, cex = tid - cey * cwin;
    const bool has1 = e1 < rwin * rwin, hasc = tid < cwin * cwin;
    constexpr int RW = QT + 4;
    const int m0y = tid / RW, m0x = tid - m0y * RW;
    const int m1 = tid + 256, m1y = m1 / RW, m1x = m1 - m1y * RW;
    const bool hasm1 = LMIN && m1 < RW * RW;
    const int moff0 = clampi(ly0 - 2 + m0y, 0, g.H - 1) * g.W + clampi(lx0 - 2 + m0x, 0, g.W - 1);
    const int moff1 = clampi(ly0 - 2 + m1y, 0, g.H - 1) * g.W + clampi(lx0 - 2 + m1x, 0, g.W - 1);
    const int nloop = a.n + ((a.flags & HHSR_MERGE_DO_REF) ? 1 : 0);
    float pr0 = 0.f, pr1 = 0.f, plr = 0.f, plr1 = 0.f;
    float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 pfl = make_float2(0.f, 0.f);
    auto is_edge = [&](int n) { return n >= a.n ? edge_ref : (bool)((edge_mask >> n) & 1ull); };  // wave-uniform
    auto prefetch = [&](int n, const bool edge_n) {
        const bool isref = n >= a.n;
        const float* __restrict__ raw = isref ? a.ref_raw : a.f[n].raw;
        const float4* __restrict__ cov = isref ? a.ref_cov : a.f[n].cov;
        int ox = lx0 - 1, oy = ly0 - 1;
        if (!isref) {
            pfl = a.f[n].flow[tile];
            ox = xs_comp_org<S>(pfl.x, lx0);
            oy = xs_comp_org<S>(pfl.y, ly0);
        }
        if (edge_n) {  // clamped coordinates; the mask says which window positions are real samples
            const int y0 = oy + e0y, x0 = ox + e0x, y1 = oy + e1y, x1 = ox + e1x;
            pr0 = raw[(size_t)clampi(y0, 0, g.H - 1) * g.pitch + clampi(x0, 0, g.W - 1)];
            if (has1) pr1 = raw[(size_t)clampi(y1, 0, g.H - 1) * g.pitch + clampi(x1, 0, g.W - 1)];
            if (!ISO && hasc)
                pc = cov[(size_t)clampi((oy >> 1) + cey, 0, g.gh - 1) * g.gw + clampi((ox >> 1) + cex, 0, g.gw - 1)];
        } else {
        pr0 = raw[(size_t)(oy + e0y) * g.pitch + ox + e0x];
        if (has1) pr1 = raw[(size_t)(oy + e1y) * g.pitch + ox + e1x];
        if (!ISO && hasc) pc = cov[(size_t)min((oy >> 1) + cey, g.gh - 1) * g.gw + min((ox >> 1) + cex, g.gw - 1)];
        }
        if (!isref) {
            if (LMIN) {
                plr = a.f[n].r[moff0];
                if (hasm1) plr1 = a.f[n].r[moff1];
            } else {
                plr = a.f[n].r[ridx];
            }
        }
    };
    const float* __restrict__ rbase = s_R + ty * X2_RP + 2 * lj;
    const int cbase = li * X2_CP + lj;

    // The WHOLE frame loop exists twice: tiles without a single EDGE frame (all but the image's perimeter and the tiles a
    // large flow pushes over the border) run round 3's loop, in which nothing of the edge handling exists; the others run
    // the copy with the per-frame (wave-uniform, run-time) edge branches.  One loop with the branches inside cost the
    // common tiles 12 % (34.4 -> 38.8 ms over the interior rows of the 48 MP x 20 burst), two copies of only the
    // sub-pixel loop behind one branch spill 140 VGPRs; two copies of the loop cost code size only.
    auto run_frames = [&](auto edge_tile_c) __attribute__((always_inline)) {
    constexpr bool EDGE_TILE = decltype(edge_tile_c)::value;
    if (nloop > 0) prefetch(0, EDGE_TILE && is_edge(0));
    for (int n = 0; n < nloop; ++n) {
        const bool isref = n >= a.n;
        __syncthreads();
        s_rawA[e0y * X2_RP + e0x] = pr0;
        if (e0x > 0) s_rawB[e0y * X2_RP + e0x - 1] = pr0;
        if (has1) {
            s_rawA[e1y * X2_RP + e1x] = pr1;
            if (e1x > 0) s_rawB[e1y * X2_RP + e1x - 1] = pr1;
        }
        if (!ISO && hasc) s_cov[cey * X2_CP + cex] = pc;
        if (LMIN && !isref) {
            s_R[m0y * X2_RP + m0x] = plr;
            if (hasm1) s_R[m1y * X2_RP + m1x] = plr1;
        }
        const bool edge = EDGE_TILE && is_edge(n);
        if (edge) {
            int ox = lx0 - 1, oy = ly0 - 1;
            if (!isref) {
                ox = xs_comp_org<S>(pfl.x, lx0);
                oy = xs_comp_org<S>(pfl.y, ly0);
            }
            const int y0 = oy + e0y, x0 = ox + e0x, y1 = oy + e1y, x1 = ox + e1x;
            const float pm0 = (y0 >= 0 && y0 < g.H && x0 >= 0 && x0 < g.W) ? 1.f : 0.f;
            const float pm1 = (y1 >= 0 && y1 < g.H && x1 >= 0 && x1 < g.W) ? 1.f : 0.f;
            s_mskA[e0y * X2_RP + e0x] = pm0;
            if (e0x > 0) s_mskB[e0y * X2_RP + e0x - 1] = pm0;
            if (has1) {
                s_mskA[e1y * X2_RP + e1x] = pm1;
                if (e1x > 0) s_mskB[e1y * X2_RP + e1x - 1] = pm1;
            }
        }
        const float2 fl = pfl;
        float local_r = isref ? 1.f : plr;
        __syncthreads();
        if (n + 1 < nloop) prefetch(n + 1, EDGE_TILE && is_edge(n + 1));
        if (LMIN && !isref) {
            float m = 3.0e38f;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float2 v01 = lds_pair(rbase + r * X2_RP), v23 = lds_pair(rbase + r * X2_RP + 2);
                const float2 v45 = lds_pair(rbase + r * X2_RP + 4);
                m = fminf(m, fminf(fminf(v01.y, v23.x), fminf(v23.y, fminf(v45.x, px ? v45.y : v01.x))));
            }
            local_r = m;
        }
        if (!isref) racc += local_r;
        if (local_r == 0.f) continue;
        const XsAxis<S> ax = isref ? xs_ref_axis<S>(lx, lx0, 0, g.scale, lj) : xs_comp_axis<S>(fl.x, lx0, px);
        const XsAxis<S> ay = isref ? xs_ref_axis<S>(ly, ly0, g.off_lr, g.scale, li) : xs_comp_axis<S>(fl.y, ly0, py);
        // software pipeline over the S x S sub-pixels: the LDS reads of sub-pixel q + 1 (4 covariance cells, 3 x 2 raw
        // pairs) are issued before the taps of sub-pixel q are evaluated — with 72 accumulators per thread only two
        // waves fit a SIMD, too few to hide the LDS latency of read -> wait -> compute per sub-pixel
        struct Sub {
            float4 c00, c01, c10, c11;
            float2 v01[3], v23[3];
        };
        auto load_sub = [&](int sa, int sb) {
            Sub t;
            if (!ISO) {
                const int ca = cbase + ay.oc[sa] * X2_CP + ax.oc[sb];
                t.c00 = lds_quad(s_cov + ca);
                t.c01 = lds_quad(s_cov + ca + 1);
                t.c10 = lds_quad(s_cov + ca + X2_CP);
                t.c11 = lds_quad(s_cov + ca + X2_CP + 1);
            }
            const int mcol = px + ax.e[sb];
            const float* __restrict__ rp = ((mcol & 1) ? s_rawB : s_rawA) + (ty + ay.e[sa]) * X2_RP + 2 * lj + (mcol & 2);
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                t.v01[di] = lds_pair(rp + di * X2_RP);
                t.v23[di] = lds_pair(rp + di * X2_RP + 2);
            }
            return t;
        };
        const bool EDGE = edge;  // (false at compile time in the common tiles' copy of the loop)
        Sub cur = load_sub(0, 0);
#pragma unroll
        for (int q = 0; q < S * S; ++q) {
            const int sa = q / S, sb = q % S;
            Sub nxt = cur;
            if (HHSR_XS_PIPE && q + 1 < S * S) nxt = load_sub((q + 1) / S, (q + 1) % S);
            float ixx = 2.f * X2_KEXP, ixy = 0.f, iyy = 2.f * X2_KEXP;
            bool finite = true;
            float r_eff = local_r;
            float gx_e = ax.f[sb], gy_e = ay.f[sa];
            if (EDGE) {
                // window centre of this sub-pixel = floor(position): c - 1 = org + t + e
                const int cj = ax.org + tx + ax.e[sb] + 1, ci = ay.org + ty + ay.e[sa] + 1;
                if (!isref) {
                    if (!((unsigned)cj < (unsigned)g.W && (unsigned)ci < (unsigned)g.H)) r_eff = 0.f;  // position outside the frame
                    // centre in column / row 0: cells 0 and 1 with the fraction (fr - 1) / 2 (the window read below took
                    // cells -1 -> 0 (clamped) and 0 with (1 + fr) / 2: move one cell on, fraction - 1)
                    if (cj == 0) gx_e -= 1.f;
                    if (ci == 0) gy_e -= 1.f;
                }
                if (!ISO && !isref && (cj == 0 || ci == 0)) {
                    const int ca = cbase + (ay.oc[sa] + (ci == 0)) * X2_CP + ax.oc[sb] + (cj == 0);
                    cur.c00 = lds_quad(s_cov + ca);
                    cur.c01 = lds_quad(s_cov + ca + 1);
                    cur.c10 = lds_quad(s_cov + ca + X2_CP);
                    cur.c11 = lds_quad(s_cov + ca + X2_CP + 1);
                }
            }
            if (!ISO) {
                const float gx = gx_e, gy = gy_e;
                const float w11 = gx * gy, w01 = gx - w11, w10 = gy - w11, w00 = (1.f - gx) - w10;
                const float cxx = fmaf(w11, cur.c11.x, fmaf(w10, cur.c10.x, fmaf(w01, cur.c01.x, w00 * cur.c00.x)));
                const float cxy = fmaf(w11, cur.c11.y, fmaf(w10, cur.c10.y, fmaf(w01, cur.c01.y, w00 * cur.c00.y)));
                const float cyy = fmaf(w11, cur.c11.w, fmaf(w10, cur.c10.w, fmaf(w01, cur.c01.w, w00 * cur.c00.w)));
                const float det = fmaf(cxx, cyy, -(cxy * cxy));
                const float s1 = __builtin_amdgcn_rcpf(det) * X2_KEXP;
                ixx = s1 * cyy;
                ixy = (-2.f * s1) * cxy;
                iyy = s1 * cxx;
                if (isref && !(fabsf(det) > 1e-10f)) {
                    ixx = X2_KEXP;
                    ixy = 0.f;
                    iyy = X2_KEXP;
                }
                const float probe = fmaf(0.f, ixx, fmaf(0.f, ixy, 0.f * iyy));
                finite = probe == probe;
            }
            const float dx0 = ax.d0[sb], dy0 = ay.d0[sa];
            const float dxs[3] = {dx0 - 1.f, dx0, dx0 + 1.f};
            float sv[2][2], sd[2][2];
            auto taps = [&](auto exact_c, auto masked_c) {
                constexpr bool EXACT = decltype(exact_c)::value, MASKED = decltype(masked_c)::value;
                const int mcol_m = px + ax.e[sb];
                const float* __restrict__ mp = ((mcol_m & 1) ? s_mskB : s_mskA) + (ty + ay.e[sa]) * X2_RP + 2 * lj + (mcol_m & 2);
#pragma unroll
                for (int di = 0; di < 3; ++di) {
                    const float c3[3] = {cur.v01[di].x, cur.v01[di].y, cur.v23[di].x};
                    float m3[3] = {1.f, 1.f, 1.f};
                    if (MASKED) {
                        const float2 ma = lds_pair(mp + di * X2_RP), mb = lds_pair(mp + di * X2_RP + 2);
                        m3[0] = ma.x; m3[1] = ma.y; m3[2] = mb.x;
                    }
                    const float dy = dy0 + (float)(di - 1);
                    const float qa = iyy * dy * dy, qb = ixy * dy;
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) {
                        const float dx = dxs[dj];
                        const float z = fmaf(fmaf(ixx, dx, qb), dx, qa);
                        float w = EXACT ? __builtin_amdgcn_exp2f(fminf(z, 0.f))
                                        : __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(z), 0.f, 1.f);
                        if (MASKED) w *= m3[dj];  // a tap outside the frame does not exist (merge.py:404-405)
                        if (di < 2 && dj < 2) {
                            sv[di & 1][dj & 1] = w * c3[dj];
                            sd[di & 1][dj & 1] = w;
                        } else {
                            sv[di & 1][dj & 1] = fmaf(w, c3[dj], sv[di & 1][dj & 1]);
                            sd[di & 1][dj & 1] += w;
                        }
                    }
                }
            };
            if (EDGE) taps(std::true_type{}, std::true_type{});  // (rare: the exact form, masked)
            else if (ISO || finite) taps(std::false_type{}, std::false_type{});
            else taps(std::true_type{}, std::false_type{});
            const int by = (ay.org + py + ay.e[sa]) & 1, bx = (ax.org + px + ax.e[sb]) & 1;
#if HHSR_XS_RGB
            // tap parity (a, b) is colour class (a ^ by, b ^ bx): red sits at parity (ri ^ by, rj ^ bx), blue diagonally
            // opposite, the greens on the other diagonal — four wave-uniform arrangements
            const int ra = ri ^ by, rb = rj ^ bx;
#define HHSR_FOLD3(RA, RB)                                                                    \
    {                                                                                         \
        nacc[sa][sb][0] = fmaf(r_eff, sv[RA][RB], nacc[sa][sb][0]);                           \
        dacc[sa][sb][0] = fmaf(r_eff, sd[RA][RB], dacc[sa][sb][0]);                           \
        nacc[sa][sb][1] = fmaf(r_eff, sv[RA ^ 1][RB] + sv[RA][RB ^ 1], nacc[sa][sb][1]);      \
        dacc[sa][sb][1] = fmaf(r_eff, sd[RA ^ 1][RB] + sd[RA][RB ^ 1], dacc[sa][sb][1]);      \
        nacc[sa][sb][2] = fmaf(r_eff, sv[RA ^ 1][RB ^ 1], nacc[sa][sb][2]);                   \
        dacc[sa][sb][2] = fmaf(r_eff, sd[RA ^ 1][RB ^ 1], dacc[sa][sb][2]);                   \
    }
            if (ra) {
                if (rb) { asm volatile("; xs fold 11"); HHSR_FOLD3(1, 1) asm volatile("; xs end 11"); }
                else { asm volatile("; xs fold 10"); HHSR_FOLD3(1, 0) asm volatile("; xs end 10"); }
            } else {
                if (rb) { asm volatile("; xs fold 01"); HHSR_FOLD3(0, 1) asm volatile("; xs end 01"); }
                else { asm volatile("; xs fold 00"); HHSR_FOLD3(0, 0) asm volatile("; xs end 00"); }
            }
#undef HHSR_FOLD3
#else
#define HHSR_FOLD(BY, BX)                                                                             \
    _Pragma("unroll") for (int aa = 0; aa < 2; ++aa) _Pragma("unroll") for (int bb = 0; bb < 2; ++bb) { \
        nacc[sa][sb][aa * 2 + bb] = fmaf(r_eff, sv[aa ^ BY][bb ^ BX], nacc[sa][sb][aa * 2 + bb]);      \
        dacc[sa][sb][aa * 2 + bb] = fmaf(r_eff, sd[aa ^ BY][bb ^ BX], dacc[sa][sb][aa * 2 + bb]);      \
    }
            if (by) {
                if (bx) { asm volatile("; xs fold 11"); HHSR_FOLD(1, 1) asm volatile("; xs end 11"); }
                else { asm volatile("; xs fold 10"); HHSR_FOLD(1, 0) asm volatile("; xs end 10"); }
            } else {
                if (bx) { asm volatile("; xs fold 01"); HHSR_FOLD(0, 1) asm volatile("; xs end 01"); }
                else { asm volatile("; xs fold 00"); HHSR_FOLD(0, 0) asm volatile("; xs end 00"); }
            }
#undef HHSR_FOLD
#endif
            if (HHSR_XS_PIPE) cur = nxt;
            else if (q + 1 < S * S) cur = load_sub((q + 1) / S, (q + 1) % S);
        }
    }
    };  // run_frames
    if (edge_mask != 0ull || ((a.flags & HHSR_MERGE_DO_REF) && edge_ref)) run_frames(std::true_type{});
    else run_frames(std::false_type{});
    if (a.acc_r) a.acc_r[ridx] = ((a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[ridx] : 0.f) + racc;
    if (a.flags & HHSR_MERGE_LOAD_ACC) {
#pragma unroll
        for (int sa = 0; sa < S; ++sa)
#pragma unroll
            for (int sb = 0; sb < S; ++sb) {
                const int hi = S * ly + sa, hj = S * lx + sb;
                if (border_pixel(g, hi, hj)) continue;
                const size_t o = ((size_t)(hi - g.row0) * g.sW + hj) * 3;
                float n3[3], d3[3];
                xs_rgb<NC>(cfa, nacc[sa][sb], dacc[sa][sb], n3, d3);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float nk = num[o + k] + n3[k], dk = den[o + k] + d3[k];
                    num[o + k] = (a.flags & HHSR_MERGE_DIVIDE) ? nk / dk : nk;
                    if (a.flags & HHSR_MERGE_STORE_DEN) den[o + k] = dk;
                }
            }
        return;
    }
    const int npass = (a.flags & HHSR_MERGE_STORE_DEN) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int sa = 0; sa < S; ++sa) {
            float* row = s_out + (S * ty + sa) * OP + 3 * S * tx;
#pragma unroll
            for (int sb = 0; sb < S; ++sb) {
                float n3[3], d3[3];
                xs_rgb<NC>(cfa, nacc[sa][sb], dacc[sa][sb], n3, d3);
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    row[3 * sb + k] = pass ? d3[k] : ((a.flags & HHSR_MERGE_DIVIDE) ? n3[k] / d3[k] : n3[k]);
            }
        }
        __syncthreads();
        float* __restrict__ dst = pass ? den : num;
        constexpr int CPR = 3 * S * QT / 4, NCH = OROWS * CPR;  // float4 chunks per tile row / per tile
        for (int qd = tid; qd < NCH; qd += 256) {
            const int orow = qd / CPR, oc = (qd - orow * CPR) * 4;
            *reinterpret_cast<float4*>(dst + ((size_t)(S * ly0 + orow - g.row0) * g.sW + S * lx0) * 3 + oc) =
                *reinterpret_cast<const float4*>(s_out + orow * OP + oc);
        }
    }
}

// ---- x3, second generation (round 4): 768-thread workgroups, wave = parity class x output SUB-ROW -------------------------
// k_merge_xs<3> carries 3 x 3 sub-pixels x 8 accumulators = 72 accumulators per thread: two waves per SIMD, and it waits on
// LDS / transcendental latency there (41.8 ms at 48 MP x 20, 66 % of k_merge_x2's per-instruction rate).  Every
// register-only variant of that design spills at three waves.  Here the tile's work is split the other
// way: the SAME 16 x 16 LR tile, staged ONCE per frame, is worked on by 12 waves — wave w owns parity class w & 3 and
// output sub-row w >> 2, a thread owns one LR pixel's three sub-pixels of that sub-row.  Per thread: 3 x 6 accumulators
// (Bayer sensors: the two green classes are summed when a frame is folded — red / green / blue instead of four
// classes), ~3 x fewer taps per frame, the same staging slots spread over 3 x the threads; the wave-uniform float64
// geometry of a frame is evaluated once per workgroup (lane = frame) into an LDS table that the waves read back with
// broadcast loads.  <= 168 VGPRs: three waves per SIMD (one 12-wave workgroup per CU).
// Non-Bayer 2 x 2 colour layouts keep k_merge_xs (the channel fold below needs red and blue on one diagonal).
#ifndef HHSR_X3W_OCC
#define HHSR_X3W_OCC 3
#endif
#ifndef HHSR_X3W
#define HHSR_X3W 0  // 1: scale 3 runs k_merge_x3w by default.  Measured (48 MP x 20 x3, tools/debug/ab_c5.sh): k_merge_xs<3>
                    // 41.3 ms (124 VGPRs + 72 accumulators, 2 waves / SIMD), k_merge_x3w 47.8 ms (147 VGPRs, 3 waves / SIMD,
                    // one 12-wave workgroup per CU), double-buffered (one barrier per frame) 48.1 ms: occupancy is not what
                    // k_merge_xs<3> lacks.  The kernel stays selectable (HHSR_MERGE_FORCE_X3W, config.hip.merge_kernel:
                    // x3w) and tested.
#endif
#ifndef HHSR_X3W_DB
#define HHSR_X3W_DB 1  // 1: double-buffered LDS windows, one barrier per frame
#endif
constexpr int X3_NT = 768;

static bool cfa_is_bayer(const Cfa4& c) {  // red (0) and blue (2) on one diagonal, green (1) on the other
    for (int k = 0; k < 4; ++k)
        if (c.c[k] == 0) return c.c[3 - k] == 2 && c.c[k ^ 1] == 1 && c.c[k ^ 2] == 1;
    return false;
}

template <bool ISO, bool LMIN>
__global__ void __launch_bounds__(X3_NT, HHSR_X3W_OCC) k_merge_x3w(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                                   float* __restrict__ den) {
    constexpr int S = 3;
    constexpr int RAWSZ = 20 * X2_RP, COVSZ = CWIN * X2_CP, OP = 3 * S * QT + 4, OROWS = S * QT;
    constexpr int GQ = 14;  // float4 per frame of the geometry table: x [parity][4], y [parity][sub-row]
    constexpr int NB = HHSR_X3W_DB ? 2 : 1;  // window buffers (2: frame n + 1 is staged while frame n is evaluated)
    __shared__ __align__(16) float s_rawA[NB * RAWSZ];
    __shared__ __align__(16) float s_rawB[NB * RAWSZ];
    __shared__ float4 s_cov[NB * COVSZ];
    __shared__ __align__(16) float s_R[NB * RAWSZ];
    __shared__ __align__(16) float s_out[OROWS * OP];
    __shared__ float4 s_geo[HHSR_MAX_FRAMES * GQ];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int cls = wave & 3, sa = wave >> 2;  // parity class and output sub-row of this wave
    const int nbx = gridDim.x, nblk = gridDim.x * gridDim.y;
    const int bid = xcd_remap(blockIdx.y * nbx + blockIdx.x, nblk);
    const int lx0 = (bid % nbx) * QT, ly0 = g.row0 / S + (bid / nbx) * QT;  // LR origin of the workgroup
    const int lrow1 = g.row1 / S;
    const int tile = (min(ly0, g.H - 1) / g.ts) * g.nx + min(lx0, g.W - 1) / g.ts;
    const int py = cls >> 1, px = cls & 1;
    const int li = lane >> 3, lj = lane & 7;
    const int ty = 2 * li + py, tx = 2 * lj + px;
    const int ly = ly0 + ty, lx = lx0 + tx;

    bool ok = lx0 + QT <= g.W && ly0 + QT <= lrow1;
    if ((a.flags & HHSR_MERGE_DO_REF) && !(lx0 >= 1 && lx0 + QT + 2 <= g.W && ly0 >= 1 && ly0 + QT + 2 <= g.H)) ok = false;
    if (ok && lane < a.n) {
        const float2 fl = a.f[lane].flow[tile];
        const int ox = xs_comp_org<S>(fl.x, lx0), oy = xs_comp_org<S>(fl.y, ly0);
        ok = ox >= 0 && ox + X2_WIN <= g.W && oy >= 0 && oy + X2_WIN <= g.H && fl.x == fl.x && fl.y == fl.y;
    }
    if (!__all(ok)) {  // (identical in the twelve waves: every wave looks at every frame)
        // generic per-pixel code from global memory for the three output pixels of this thread
        if (lx >= g.W || ly >= lrow1) return;
        if (a.acc_r && sa == 0) {
            float racc = (a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[(size_t)ly * g.W + lx] : 0.f;
            for (int n = 0; n < a.n; ++n) racc += robustness_at(a.f[n].r, g, ly, lx, LMIN);
            a.acc_r[(size_t)ly * g.W + lx] = racc;
        }
#pragma unroll 1
        for (int sb = 0; sb < S; ++sb) {
            const int hi = S * ly + sa, hj = S * lx + sb;
            if (!border_pixel(g, hi, hj)) merge_pixel<float, GEOM_F64, ISO>(a, g, cfa, hi, hj, num, den);
        }
        return;
    }

    // per-frame geometry, once per workgroup: lane = frame (visible after the first barrier of the frame loop)
    if (tid < a.n) {
        const float2 fl = a.f[tid].flow[tile];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const XsAxis<S> u = xs_comp_axis<S>(fl.x, lx0, p);
            float4* q = s_geo + tid * GQ + p * 4;
            q[0] = make_float4(__int_as_float(u.org), __int_as_float(u.e[0]), __int_as_float(u.e[1]), __int_as_float(u.e[2]));
            q[1] = make_float4(u.d0[0], u.d0[1], u.d0[2], 0.f);
            q[2] = make_float4(__int_as_float(u.oc[0]), __int_as_float(u.oc[1]), __int_as_float(u.oc[2]), 0.f);
            q[3] = make_float4(u.f[0], u.f[1], u.f[2], 0.f);
            const XsAxis<S> v = xs_comp_axis<S>(fl.y, ly0, p);
#pragma unroll
            for (int k = 0; k < S; ++k)
                s_geo[tid * GQ + 8 + p * 3 + k] = make_float4(__int_as_float(v.org), __int_as_float(v.e[k] | (v.oc[k] << 8)), v.d0[k], v.f[k]);
        }
    }

    const int ridx = ly * g.W + lx;
    float n3[S][3], d3[S][3];
#pragma unroll
    for (int k = 0; k < S * 3; ++k) {
        (&n3[0][0])[k] = 0.f;
        (&d3[0][0])[k] = 0.f;
    }
    float racc = 0.f;
    // class index of the red (channel 0) sample; blue is the other end of that diagonal, the two greens the other diagonal
    const int rcl = cfa.c[0] == 0 ? 0 : cfa.c[1] == 0 ? 1 : cfa.c[2] == 0 ? 2 : 3;
    const int ri = rcl >> 1, rj = rcl & 1;

    // staging slots, by thread id: raw window [0, 361), covariance cells [384, 505), R window [512, 768) + [0, 144)
    constexpr int rwin = X2_WIN, cwin = QT / 2 + 3, RW = QT + 4;
    const bool hasr = tid < rwin * rwin;
    const int e0y = tid / rwin, e0x = tid - e0y * rwin;
    const int ct = tid - 384;
    const bool hasc = ct >= 0 && ct < cwin * cwin;
    const int cey = max(ct, 0) / cwin, cex = max(ct, 0) - cey * cwin;
    const int m0 = tid >= 512 ? tid - 512 : tid + 256;
    const bool hasm = LMIN && (tid >= 512 || tid < RW * RW - 256);
    const int m0y = m0 / RW, m0x = m0 - m0y * RW;
    const int moff0 = clampi(ly0 - 2 + m0y, 0, g.H - 1) * g.W + clampi(lx0 - 2 + m0x, 0, g.W - 1);
    const int nloop = a.n + ((a.flags & HHSR_MERGE_DO_REF) ? 1 : 0);
    float pr0 = 0.f, plr = 0.f;
    float4 pc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto prefetch = [&](int n) {
        const bool isref = n >= a.n;
        const float* __restrict__ raw = isref ? a.ref_raw : a.f[n].raw;
        const float4* __restrict__ cov = isref ? a.ref_cov : a.f[n].cov;
        int ox = lx0 - 1, oy = ly0 - 1;
        if (!isref) {
            const float2 fl = a.f[n].flow[tile];
            ox = xs_comp_org<S>(fl.x, lx0);
            oy = xs_comp_org<S>(fl.y, ly0);
        }
        if (hasr) pr0 = raw[(size_t)(oy + e0y) * g.pitch + ox + e0x];
        if (!ISO && hasc) pc = cov[(size_t)min((oy >> 1) + cey, g.gh - 1) * g.gw + min((ox >> 1) + cex, g.gw - 1)];
        if (!isref) {
            if (LMIN) {
                if (hasm) plr = a.f[n].r[moff0];
            } else {
                plr = a.f[n].r[ridx];
            }
        }
    };
    const float* __restrict__ rbase = s_R + ty * X2_RP + 2 * lj;
    const int cbase = li * X2_CP + lj;

    // stage(n, bo): the prefetched registers of frame n -> window buffer bo; sr = that frame's own robustness value
    float sr = 0.f;
    auto stage = [&](int n, int bo) {
        const bool isref = n >= a.n;
        if (hasr) {
            s_rawA[bo * RAWSZ + e0y * X2_RP + e0x] = pr0;
            if (e0x > 0) s_rawB[bo * RAWSZ + e0y * X2_RP + e0x - 1] = pr0;
        }
        if (!ISO && hasc) s_cov[bo * COVSZ + cey * X2_CP + cex] = pc;
        if (LMIN && !isref && hasm) s_R[bo * RAWSZ + m0y * X2_RP + m0x] = plr;
        sr = isref ? 1.f : plr;
    };
#if HHSR_X3W_DB
    // ONE workgroup barrier per frame: with a single 12-wave workgroup per CU nobody fills the time a barrier costs
    if (nloop > 0) {
        prefetch(0);
        __syncthreads();  // (the geometry table)
        stage(0, 0);
        __syncthreads();
        if (nloop > 1) prefetch(1);
    }
#else
    if (nloop > 0) prefetch(0);
#endif
    for (int n = 0; n < nloop; ++n) {
        const bool isref = n >= a.n;
#if HHSR_X3W_DB
        const int bo = n & 1;
        float local_r = sr;
#else
        const int bo = 0;
        __syncthreads();
        stage(n, 0);
        float local_r = sr;
        __syncthreads();
        if (n + 1 < nloop) prefetch(n + 1);
#endif
        if (LMIN && !isref) {  // 5 x 5 minimum of the robustness window (bit patterns: R is in [0, 1], see k_merge_x2)
            unsigned m = 0x7f7fffffu;
#pragma unroll
            for (int r = 0; r < 5; ++r) {
                const float2 v01 = lds_pair(rbase + bo * RAWSZ + r * X2_RP), v23 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 2);
                const float2 v45 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 4);
                m = min(m, min(min(__float_as_uint(v01.y), __float_as_uint(v23.x)),
                               min(__float_as_uint(v23.y), min(__float_as_uint(v45.x), __float_as_uint(px ? v45.y : v01.x)))));
            }
            local_r = __uint_as_float(m);
        }
        if (!isref && sa == 0) racc += local_r;
        if (local_r != 0.f) {
        XsAxis<S> ax;
        int ay_org, ay_e, ay_oc;
        float ay_d0, ay_f;
        if (isref) {  // the reference frame's positions are float32 per LR pixel (merge.py:113-114): per-thread geometry
            ax = xs_ref_axis<S>(lx, lx0, 0, g.scale, lj);
            const XsAxis<S> ay = xs_ref_axis<S>(ly, ly0, g.off_lr, g.scale, li);
            ay_org = ay.org;
            ay_e = sa == 0 ? ay.e[0] : sa == 1 ? ay.e[1] : ay.e[2];
            ay_oc = sa == 0 ? ay.oc[0] : sa == 1 ? ay.oc[1] : ay.oc[2];
            ay_d0 = sa == 0 ? ay.d0[0] : sa == 1 ? ay.d0[1] : ay.d0[2];
            ay_f = sa == 0 ? ay.f[0] : sa == 1 ? ay.f[1] : ay.f[2];
        } else {
            const float4* q = s_geo + n * GQ + px * 4;
            const float4 q0 = lds_quad(q), q1 = lds_quad(q + 1), q2 = lds_quad(q + 2), q3 = lds_quad(q + 3);
            const float4 qy = lds_quad(s_geo + n * GQ + 8 + py * 3 + sa);
            ax.org = __float_as_int(q0.x);
            ax.e[0] = __float_as_int(q0.y); ax.e[1] = __float_as_int(q0.z); ax.e[2] = __float_as_int(q0.w);
            ax.d0[0] = q1.x; ax.d0[1] = q1.y; ax.d0[2] = q1.z;
            ax.oc[0] = __float_as_int(q2.x); ax.oc[1] = __float_as_int(q2.y); ax.oc[2] = __float_as_int(q2.z);
            ax.f[0] = q3.x; ax.f[1] = q3.y; ax.f[2] = q3.z;
            ay_org = __float_as_int(qy.x);
            const int pk = __float_as_int(qy.y);
            ay_e = pk & 0xff;
            ay_oc = pk >> 8;
            ay_d0 = qy.z;
            ay_f = qy.w;
        }
        struct Sub {
            float4 c00, c01, c10, c11;
            float2 v01[3], v23[3];
        };
        auto load_sub = [&](int sb) {
            Sub t;
            if (!ISO) {
                const int ca = bo * COVSZ + cbase + ay_oc * X2_CP + ax.oc[sb];
                t.c00 = lds_quad(s_cov + ca);
                t.c01 = lds_quad(s_cov + ca + 1);
                t.c10 = lds_quad(s_cov + ca + X2_CP);
                t.c11 = lds_quad(s_cov + ca + X2_CP + 1);
            }
            const int mcol = px + ax.e[sb];
            const float* __restrict__ rp = ((mcol & 1) ? s_rawB : s_rawA) + bo * RAWSZ + (ty + ay_e) * X2_RP + 2 * lj + (mcol & 2);
#pragma unroll
            for (int di = 0; di < 3; ++di) {
                t.v01[di] = lds_pair(rp + di * X2_RP);
                t.v23[di] = lds_pair(rp + di * X2_RP + 2);
            }
            return t;
        };
        const int by = (ay_org + py + ay_e) & 1;
        Sub cur = load_sub(0);
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            Sub nxt = cur;
            if (sb + 1 < S) nxt = load_sub(sb + 1);
            float ixx = 2.f * X2_KEXP, ixy = 0.f, iyy = 2.f * X2_KEXP;
            bool finite = true;
            if (!ISO) {
                const float gx = ax.f[sb], gy = ay_f;
                const float w11 = gx * gy, w01 = gx - w11, w10 = gy - w11, w00 = (1.f - gx) - w10;
                const float cxx = fmaf(w11, cur.c11.x, fmaf(w10, cur.c10.x, fmaf(w01, cur.c01.x, w00 * cur.c00.x)));
                const float cxy = fmaf(w11, cur.c11.y, fmaf(w10, cur.c10.y, fmaf(w01, cur.c01.y, w00 * cur.c00.y)));
                const float cyy = fmaf(w11, cur.c11.w, fmaf(w10, cur.c10.w, fmaf(w01, cur.c01.w, w00 * cur.c00.w)));
                const float det = fmaf(cxx, cyy, -(cxy * cxy));
                const float s1 = __builtin_amdgcn_rcpf(det) * X2_KEXP;
                ixx = s1 * cyy;
                ixy = (-2.f * s1) * cxy;
                iyy = s1 * cxx;
                if (isref && !(fabsf(det) > 1e-10f)) {
                    ixx = X2_KEXP;
                    ixy = 0.f;
                    iyy = X2_KEXP;
                }
                const float probe = fmaf(0.f, ixx, fmaf(0.f, ixy, 0.f * iyy));
                finite = probe == probe;
            }
            const float dx0 = ax.d0[sb], dy0 = ay_d0;
            const float dxs[3] = {dx0 - 1.f, dx0, dx0 + 1.f};
            float sv[2][2], sd[2][2];
            auto taps = [&](auto exact_c) {
                constexpr bool EXACT = decltype(exact_c)::value;
#pragma unroll
                for (int di = 0; di < 3; ++di) {
                    const float c3[3] = {cur.v01[di].x, cur.v01[di].y, cur.v23[di].x};
                    const float dy = dy0 + (float)(di - 1);
                    const float qa = iyy * dy * dy, qb = ixy * dy;
#pragma unroll
                    for (int dj = 0; dj < 3; ++dj) {
                        const float dx = dxs[dj];
                        const float z = fmaf(fmaf(ixx, dx, qb), dx, qa);
                        const float w = EXACT ? __builtin_amdgcn_exp2f(fminf(z, 0.f))
                                              : __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(z), 0.f, 1.f);
                        if (di < 2 && dj < 2) {
                            sv[di & 1][dj & 1] = w * c3[dj];
                            sd[di & 1][dj & 1] = w;
                        } else {
                            sv[di & 1][dj & 1] = fmaf(w, c3[dj], sv[di & 1][dj & 1]);
                            sd[di & 1][dj & 1] += w;
                        }
                    }
                }
            };
            if (ISO || finite) taps(std::false_type{});
            else taps(std::true_type{});
            // tap parity (a, b) is colour class (a ^ by, b ^ bx): red sits at parity (ri ^ by, rj ^ bx), blue diagonally
            // opposite, the greens on the other diagonal — four wave-uniform arrangements
            const int bx = (ax.org + px + ax.e[sb]) & 1;
            const int ra = ri ^ by, rb = rj ^ bx;
#define HHSR_FOLD3(RA, RB)                                                                    \
    {                                                                                         \
        n3[sb][0] = fmaf(local_r, sv[RA][RB], n3[sb][0]);                                     \
        d3[sb][0] = fmaf(local_r, sd[RA][RB], d3[sb][0]);                                     \
        n3[sb][1] = fmaf(local_r, sv[RA ^ 1][RB] + sv[RA][RB ^ 1], n3[sb][1]);                \
        d3[sb][1] = fmaf(local_r, sd[RA ^ 1][RB] + sd[RA][RB ^ 1], d3[sb][1]);                \
        n3[sb][2] = fmaf(local_r, sv[RA ^ 1][RB ^ 1], n3[sb][2]);                             \
        d3[sb][2] = fmaf(local_r, sd[RA ^ 1][RB ^ 1], d3[sb][2]);                             \
    }
            if (ra) {
                if (rb) HHSR_FOLD3(1, 1) else HHSR_FOLD3(1, 0)
            } else {
                if (rb) HHSR_FOLD3(0, 1) else HHSR_FOLD3(0, 0)
            }
#undef HHSR_FOLD3
            cur = nxt;
        }
        }  // local_r != 0
#if HHSR_X3W_DB
        // stage frame n + 1 into the other buffer (its registers were prefetched during frame n - 1's taps), ONE barrier,
        // then start the loads of frame n + 2
        if (n + 1 < nloop) stage(n + 1, (n + 1) & 1);
        __syncthreads();
        if (n + 2 < nloop) prefetch(n + 2);
#endif
    }
    if (a.acc_r && sa == 0) a.acc_r[ridx] = ((a.flags & HHSR_MERGE_LOAD_ACC) ? a.acc_r[ridx] : 0.f) + racc;
    if (a.flags & HHSR_MERGE_LOAD_ACC) {
#pragma unroll
        for (int sb = 0; sb < S; ++sb) {
            const int hi = S * ly + sa, hj = S * lx + sb;
            if (border_pixel(g, hi, hj)) continue;
            const size_t o = ((size_t)(hi - g.row0) * g.sW + hj) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float nk = num[o + k] + n3[sb][k], dk = den[o + k] + d3[sb][k];
                num[o + k] = (a.flags & HHSR_MERGE_DIVIDE) ? nk / dk : nk;
                if (a.flags & HHSR_MERGE_STORE_DEN) den[o + k] = dk;
            }
        }
        return;
    }
    const int npass = (a.flags & HHSR_MERGE_STORE_DEN) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass) __syncthreads();
        float* row = s_out + (S * ty + sa) * OP + 3 * S * tx;
#pragma unroll
        for (int sb = 0; sb < S; ++sb)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                row[3 * sb + k] = pass ? d3[sb][k] : ((a.flags & HHSR_MERGE_DIVIDE) ? n3[sb][k] / d3[sb][k] : n3[sb][k]);
        __syncthreads();
        float* __restrict__ dst = pass ? den : num;
        constexpr int CPR = 3 * S * QT / 4, NCH = OROWS * CPR;  // float4 chunks per tile row / per tile
        for (int qd = tid; qd < NCH; qd += X3_NT) {
            const int orow = qd / CPR, oc = (qd - orow * CPR) * 4;
            *reinterpret_cast<float4*>(dst + ((size_t)(S * ly0 + orow - g.row0) * g.sW + S * lx0) * 3 + oc) =
                *reinterpret_cast<const float4*>(s_out + orow * OP + oc);
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
                           max_multiplier, max_frame_count, num, den);
    else
        hipLaunchKernelGGL((k_accumulate_ref<false>), grid, block, 0, s, raw, cv, g, c, acc_rob, rad_max,
                           max_multiplier, max_frame_count, num, den);
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
    if (chained && HHSR_X2_DB) {  // (the double-buffered A/B variant's frame loop starts at frame 0: it would count the
        hhsr_set_error("hhsr_merge_burst_chain: not available in the HHSR_X2_DB variant build");  // parked frames twice)
        return -3;
    }
    if (chained && !(quad && !x2_v1 && aligned16 && !mono && (cfa_is_bayer(c) || !HHSR_X2_RGB))) {
        hhsr_set_error("hhsr_merge_burst_chain: needs the wave-per-class x2 kernel (scale 2, ts %% 16 == 0, sH = 2 H, "
                       "sW = 2 W, 16-byte aligned output, float32 weights, Bayer)");
        return -3;
    }
    if (lmin && !quad && !x3) {
        hhsr_set_error("hhsr_merge_burst: HHSR_MERGE_LOCAL_MIN needs the x2 kernel (scale 2, ts %% 16 == 0, "
                       "sH = 2 H, sW = 2 W, row0 %% 32 == 0, float32 weights)");
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
    } else if (mono) {
        const dim3 qgrid(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 2, QT));
        if (lmin) {
            if (iso) hipLaunchKernelGGL((k_merge_burst_quad<true, true, true>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_burst_quad<false, true, true>), qgrid, block, 0, s, a, g, c, num, den);
        } else {
            if (iso) hipLaunchKernelGGL((k_merge_burst_quad<true, false, true>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_burst_quad<false, false, true>), qgrid, block, 0, s, a, g, c, num, den);
        }
    } else if (quad && !x2_v1 && aligned16 && (cfa_is_bayer(c) || !HHSR_X2_RGB)) {
        const dim3 qgrid(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 2, QT));
if (lmin) {
            if (iso) hipLaunchKernelGGL((k_merge_x2<true, true>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_x2<false, true>), qgrid, block, 0, s, a, g, c, num, den);
        } else {
            if (iso) hipLaunchKernelGGL((k_merge_x2<true, false>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_x2<false, false>), qgrid, block, 0, s, a, g, c, num, den);
        }
    } else if (quad) {
        const dim3 qgrid(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 2, QT));
        if (lmin) {
            if (iso) hipLaunchKernelGGL((k_merge_burst_quad<true, true>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_burst_quad<false, true>), qgrid, block, 0, s, a, g, c, num, den);
        } else {
            if (iso) hipLaunchKernelGGL((k_merge_burst_quad<true, false>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_burst_quad<false, false>), qgrid, block, 0, s, a, g, c, num, den);
        }
    } else if (x3 && (HHSR_X3W || (force & HHSR_MERGE_FORCE_X3W)) && cfa_is_bayer(c) && !(force & HHSR_MERGE_FORCE_X2V1)) {
        // x3, Bayer: 768-thread workgroups, wave = parity class x output sub-row (k_merge_x3w)
        const dim3 qgrid(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 3, QT)), wblock(X3_NT);
        if (lmin) {
            if (iso) hipLaunchKernelGGL((k_merge_x3w<true, true>), qgrid, wblock, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_x3w<false, true>), qgrid, wblock, 0, s, a, g, c, num, den);
        } else {
            if (iso) hipLaunchKernelGGL((k_merge_x3w<true, false>), qgrid, wblock, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_x3w<false, false>), qgrid, wblock, 0, s, a, g, c, num, den);
        }
    } else if (x3) {
        // x3: the wave-per-parity-class kernel generalised to S x S sub-pixels (k_merge_xs): non-Bayer colour layouts,
        // config.hip.merge_kernel = x2_v1 (validation: the round-3 kernel)
        const dim3 qgrid(hhsr_cdiv(W, QT), hhsr_cdiv(nrows / 3, QT));
        if (lmin) {
            if (iso) hipLaunchKernelGGL((k_merge_xs<3, true, true>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_xs<3, false, true>), qgrid, block, 0, s, a, g, c, num, den);
        } else {
            if (iso) hipLaunchKernelGGL((k_merge_xs<3, true, false>), qgrid, block, 0, s, a, g, c, num, den);
            else hipLaunchKernelGGL((k_merge_xs<3, false, false>), qgrid, block, 0, s, a, g, c, num, den);
        }
    } else if (tiled) {
        const dim3 tgrid(hhsr_cdiv(sW, MT), hhsr_cdiv(nrows, MT));
#define HHSR_MT(GEOM, ISO) hipLaunchKernelGGL((k_merge_burst_tile<GEOM, ISO>), tgrid, block, 0, s, a, g, c, num, den)
        if (p2) { if (iso) HHSR_MT(GEOM_P2, true); else HHSR_MT(GEOM_P2, false); }
        else { if (iso) HHSR_MT(GEOM_F64, true); else HHSR_MT(GEOM_F64, false); }
#undef HHSR_MT
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

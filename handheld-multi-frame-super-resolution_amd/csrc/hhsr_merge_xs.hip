// k_merge_xs<S>: the wave-per-parity-class merge kernel with S x S sub-pixels per thread; instantiated for x3 (see
// hhsr_merge.h for the family overview).
#include "hhsr_merge.h"

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
    __shared__ float4 s_frm[HHSR_MAX_FRAMES];      // per comp frame: window origin (x, y as int bits), flow vector
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
            // the frame loop's staging takes the origin from here (lane = frame; identical in the four waves) instead of
            // re-deriving it per thread and frame behind a dependent load of the flow vector (k_merge_x2, round 6)
            if (wave == 0) s_frm[lane] = make_float4(__int_as_float(ox), __int_as_float(oy), fl.x, fl.y);
        }
    }
    __syncthreads();
    const unsigned long long edge_mask = __ballot(edge_f);  // bit n: frame n is an EDGE frame (identical in the four waves)
    const bool edge_ref = !(lx0 >= 1 && lx0 + QT + 2 <= g.W && ly0 >= 1 && ly0 + QT + 2 <= g.H);
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
    // Bayer sensors (the only layouts this kernel is launched for, cfa_is_bayer): the two green parity classes are summed
    // when a frame is folded, so a sub-pixel has 3 + 3 accumulators instead of 4 + 4 (54 instead of 72 per thread)
    constexpr int NC = 3;
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
    auto is_edge = [&](int n) { return n >= a.n ? edge_ref : (bool)((edge_mask >> n) & 1ull); };  // wave-uniform
    // frame-independent byte offsets of the staging slots (32 bits: one v_add_u32 + a global_load ... s[base:base+1] per load)
    const unsigned t0b = (unsigned)(e0y * g.pitch + e0x) * 4u, t1b = (unsigned)(e1y * g.pitch + e1x) * 4u;
    const unsigned mb0 = (unsigned)moff0 * 4u, mb1 = (unsigned)moff1 * 4u, rb = (unsigned)ridx * 4u;
    auto ldf = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    auto frame_org = [&](int n, int& ox, int& oy) {  // wave-uniform, in SGPRs
        ox = lx0 - 1, oy = ly0 - 1;
        if (n < a.n) {
            ox = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(s_frm)[n * 4]);
            oy = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(s_frm)[n * 4 + 1]);
        }
    };
    auto prefetch = [&](int n, const bool edge_n) {
        const bool isref = n >= a.n;
        const float* __restrict__ raw = isref ? a.ref_raw : a.f[n].raw;
        const float4* __restrict__ cov = isref ? a.ref_cov : a.f[n].cov;
        int ox, oy;
        frame_org(n, ox, oy);
        if (edge_n) {  // clamped coordinates; the mask says which window positions are real samples
            const int y0 = oy + e0y, x0 = ox + e0x, y1 = oy + e1y, x1 = ox + e1x;
            pr0 = raw[(size_t)clampi(y0, 0, g.H - 1) * g.pitch + clampi(x0, 0, g.W - 1)];
            if (has1) pr1 = raw[(size_t)clampi(y1, 0, g.H - 1) * g.pitch + clampi(x1, 0, g.W - 1)];
            if (!ISO && hasc)
                pc = cov[(size_t)clampi((oy >> 1) + cey, 0, g.gh - 1) * g.gw + clampi((ox >> 1) + cex, 0, g.gw - 1)];
        } else {
            const unsigned ob = (unsigned)(oy * g.pitch + ox) * 4u;  // (scalar)
            pr0 = ldf(raw, ob + t0b);
            if (has1) pr1 = ldf(raw, ob + t1b);
            if (!ISO && hasc) {
                const unsigned ci = (unsigned)(min((oy >> 1) + cey, g.gh - 1) * g.gw + min((ox >> 1) + cex, g.gw - 1));
                pc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cov) + ci * 16u);
            }
        }
        if (!isref) {
            if (LMIN) {
                plr = ldf(a.f[n].r, mb0);
                if (hasm1) plr1 = ldf(a.f[n].r, mb1);
            } else {
                plr = ldf(a.f[n].r, rb);
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
        int isref_s = __builtin_amdgcn_readfirstlane((int)(n >= a.n));  // the flag as an opaque scalar integer (see k_merge_x2)
        asm volatile("" : "+s"(isref_s));
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
            int ox, oy;
            frame_org(n, ox, oy);
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
        float2 fl = make_float2(0.f, 0.f);  // (wave-uniform: a broadcast LDS read)
        if (!isref) fl = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(s_frm) + n * 4 + 2);
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
            if (q + 1 < S * S) nxt = load_sub((q + 1) / S, (q + 1) % S);
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
                if (isref_s) {  // wave-uniform; a real branch (see k_merge_x2)
                    asm volatile("; ref identity");
                    if (!(fabsf(det) > 1e-10f)) {
                        ixx = X2_KEXP;
                        ixy = 0.f;
                        iyy = X2_KEXP;
                    }
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
                    const float dy = di == 1 ? dy0 : dy0 + (float)(di - 1);
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
            cur = nxt;
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


void hhsr_launch_merge_x3(bool iso, bool lmin, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g, const Cfa4& c,
                          float* num, float* den) {
    const dim3 block(256);
    if (lmin) {
        if (iso) hipLaunchKernelGGL((k_merge_xs<3, true, true>), grid, block, 0, s, a, g, c, num, den);
        else hipLaunchKernelGGL((k_merge_xs<3, false, true>), grid, block, 0, s, a, g, c, num, den);
    } else {
        if (iso) hipLaunchKernelGGL((k_merge_xs<3, true, false>), grid, block, 0, s, a, g, c, num, den);
        else hipLaunchKernelGGL((k_merge_xs<3, false, false>), grid, block, 0, s, a, g, c, num, den);
    }
}

// k_merge_x2: the fused burst merge at x2, one wave per Bayer parity class (see hhsr_merge.h for the family overview and
// the comment block above the HHSR_X2_* switches there for the design).
#include "hhsr_merge.h"

// Instruction budget (tools/isa_budget.py): the `//@ name` tags below open the source regions the script attributes the
// kernel's instructions to (through the line table of a -gline-tables-only build).  -DHHSR_X2_BUDGET (analysis builds only)
// leaves out the tap arm for non-finite coefficients, which ordinary data never executes.

template <bool ISO, bool LMIN>
__global__ void __launch_bounds__(256, HHSR_X2_OCC) k_merge_x2(BurstArgs a, Geo g, Cfa4 cfa, float* __restrict__ num,
                                                   float* __restrict__ den) {
    constexpr int NB = 1;  // window buffers (a double-buffered variant with one barrier per frame measured slower: 3.76 vs 3.64 ms, round 2)
    constexpr int RAWSZ = 20 * X2_RP, COVSZ = CWIN * X2_CP;
    __shared__ __align__(16) float s_rawA[NB * RAWSZ];              // window[y][x]
    __shared__ __align__(16) float s_rawB[NB * RAWSZ];              // window[y][x + 1]
    __shared__ float4 s_cov[NB * COVSZ];
    __shared__ __align__(16) float s_R[NB * RAWSZ];                 // LMIN: un-filtered robustness, tile + 2-pixel border
    __shared__ __align__(16) float s_out[32 * X2_OP];
    __shared__ float4 s_geo[(HHSR_MAX_FRAMES + 1) * 8];             // per frame: [axis x, y][parity 0, 1] x 2 quads
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

    // Per-frame geometry once per WORKGROUP: it only depends on the frame's flow vector and the parity class, so
    // evaluating it in every thread and frame (~50 instructions, ~40 % of them half-rate, identical in all lanes of a
    // wave) was 7 % of the kernel's VALU time.  Lane = frame here; the frame loop reads its entry back with four
    // broadcast ds_read_b128.  
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
                // (the y axis' cell offsets are stored times the cell window's pitch: the frame loop adds them as they are)
                const int ocs = axis ? X2_CP : 1;
                s_geo[n * 8 + axis * 4 + p * 2 + 1] = make_float4(__int_as_float(u.oc[0] * ocs), __int_as_float(u.oc[1] * ocs), u.f[0], u.f[1]);
            }
        }
    }
    __syncthreads();  // (the first prefetch below reads the table)
    const int py = wave >> 1, px = wave & 1;                        // this wave's parity class
    const int li = lane >> 3, lj = lane & 7;
    const int ty = 2 * li + py, tx = 2 * lj + px;                   // LR pixel inside the tile
    const int ridx = (ly0 + ty) * g.W + lx0 + tx;
    // Bayer layouts (the only ones this kernel is launched for): the two green parity classes are summed when a frame is
    // folded: 3 + 3 accumulators per sub-pixel instead of 4 + 4 (24 instead of 32 per thread), and the epilogue has no
    // class -> channel step (whose private arrays lived in scratch; round 4: 3.37 -> 3.24 ms).
    constexpr int NC = 3, NA = 4 * NC;
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
    // Byte offsets of the staging slots that do not depend on the frame; the frame's window origin is wave-uniform (read
    // back from the geometry table into SGPRs), so a load is ONE v_add_u32 + global_load ... s[base:base+1] — the first
    // version recomputed floor / compare / convert of the flow vector per thread and frame behind a dependent global load of
    // that vector, and every address in 64 bits (v_mad_i64_i32, v_lshl_add_u64): 39 VALU instructions per thread and frame.
    const unsigned cinv = (unsigned)(cey * g.gw + cex);
    const unsigned t0b = (unsigned)(e0y * g.pitch + e0x) * 4u, t1b = (unsigned)(e1y * g.pitch + e1x) * 4u;
    const unsigned mb0 = (unsigned)moff0 * 4u, mb1 = (unsigned)moff1 * 4u, rb = (unsigned)ridx * 4u;
    auto ldf = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
    auto prefetch = [&](int n) {  // all windows are inside the image (checked above): no bounds tests
        //@ prefetch
        const bool isref = n >= a.n;
        const float* __restrict__ raw = isref ? a.ref_raw : a.f[n].raw;
        const float4* __restrict__ cov = isref ? a.ref_cov : a.f[n].cov;
        // (the table's org of either parity: x2_comp_org of the frame's flow, l0 - 1 for the reference frame)
        const int ox = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(s_geo)[(n * 8) * 4]);
        const int oy = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(s_geo)[(n * 8 + 4) * 4]);
        const unsigned ob = (unsigned)(oy * g.pitch + ox) * 4u;  // (scalar)
        pr0 = ldf(raw, ob + t0b);
        if (has1) pr1 = ldf(raw, ob + t1b);
        if (!ISO && hasc) {
            // cell window origin (ox >> 1, oy >> 1); cells beyond the last row / column repeat it (the reference's clamp).
            // Windows that stay inside the cell grid — all but the last tile row / column's — take ONE v_add per thread.
            const int cx0 = ox >> 1, cy0 = oy >> 1;
            unsigned ci;
            if (cx0 + cwin <= g.gw && cy0 + cwin <= g.gh) {  // (scalar)
                ci = (unsigned)(cy0 * g.gw + cx0) + cinv;
            } else {
                asm volatile("; clamped cells");  // (keeps the arm a branch)
                ci = (unsigned)(min(cy0 + cey, g.gh - 1) * g.gw + min(cx0 + cex, g.gw - 1));
            }
            pc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(cov) + ci * 16u);
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

    //@ outside
    const float* __restrict__ rbase = s_R + ty * X2_RP + 2 * lj;
    const int cbase = li * X2_CP + lj;
    const int tyrp = ty * X2_RP + 2 * lj;

    // write the prefetched registers of one frame into window buffer `bo`; returns that frame's flow / robustness
    float2 sfl = make_float2(0.f, 0.f);
    float sr = 0.f;
    //@ outside
    auto stage = [&](int n, int bo) {
        //@ staging
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
    //@ outside
    auto frame = [&](auto isref_c, const bool isref_rt, const float2 fl, float local_r, const int bo, const int n) {
        const bool isref = isref_rt;  // (as a compile-time variant of the frame code: 160 VGPRs, 3.97 vs 3.54 ms — round 2)
        // the same flag as an opaque scalar INTEGER: as a bool the compiler carries it between blocks as a lane mask and
        // rebuilds the branch condition with v_cndmask + v_cmp per use
        int isref_s = __builtin_amdgcn_readfirstlane(n) - __builtin_amdgcn_readfirstlane(a.n);  // >= 0: the reference frame
        asm volatile("" : "+s"(isref_s));
        //@ min5x5
        if (LMIN && !isref) {  // 5 x 5 minimum over rows ty .. ty + 4, columns tx .. tx + 4 of the R window
            // R is clamped to [0, 1] (never negative, never NaN): the order of its float32 bit patterns is the order of
            // the values, and v_min3_u32 needs no canonicalisation of its inputs (fminf costs a v_max per operand: 28
            // half-rate instructions per thread and frame here)
            unsigned m = 0u;  // (row 0 initialises it: a constant start value is one more v_min)
            if (px) {  // (uniform branch instead of a select per row)
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const float2 v01 = lds_pair(rbase + bo * RAWSZ + r * X2_RP), v23 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 2);
                    const float2 v45 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 4);
                    const unsigned rm = min(min(__float_as_uint(v01.y), __float_as_uint(v23.x)),
                                   min(__float_as_uint(v23.y), min(__float_as_uint(v45.x), __float_as_uint(v45.y))));
                    m = r ? min(m, rm) : rm;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    const float2 v01 = lds_pair(rbase + bo * RAWSZ + r * X2_RP), v23 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 2);
                    const float2 v45 = lds_pair(rbase + bo * RAWSZ + r * X2_RP + 4);
                    const unsigned rm = min(min(__float_as_uint(v01.y), __float_as_uint(v23.x)),
                                   min(__float_as_uint(v23.y), min(__float_as_uint(v45.x), __float_as_uint(v01.x))));
                    m = r ? min(m, rm) : rm;
                }
            }
            local_r = __uint_as_float(m);
        }
        //@ geometry
        if (!isref) racc += local_r;
        if (local_r == 0.f) return;
        X2Axis ax, ay;
        {
            const float4 xa = lds_quad(s_geo + n * 8 + px * 2), xb = lds_quad(s_geo + n * 8 + px * 2 + 1);
            const float4 ya = lds_quad(s_geo + n * 8 + 4 + py * 2), yb = lds_quad(s_geo + n * 8 + 4 + py * 2 + 1);
            ax.org = __float_as_int(xa.x); ax.e[0] = 0; ax.e[1] = __float_as_int(xa.y); ax.d0[0] = xa.z; ax.d0[1] = xa.w;
            ax.oc[0] = __float_as_int(xb.x); ax.oc[1] = __float_as_int(xb.y); ax.f[0] = xb.z; ax.f[1] = xb.w;
            ay.org = __float_as_int(ya.x); ay.e[0] = 0; ay.e[1] = __float_as_int(ya.y); ay.d0[0] = ya.z; ay.d0[1] = ya.w;
            ay.oc[0] = __float_as_int(yb.x); ay.oc[1] = __float_as_int(yb.y); ay.f[0] = yb.z; ay.f[1] = yb.w;
        }
#pragma unroll
        for (int sa = 0; sa < 2; ++sa)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                float ixx = 2.f * X2_KEXP, ixy = 0.f, iyy = 2.f * X2_KEXP;
                bool finite = true;
                //@ cov_blend
                if (!ISO) {
                    const int ca = cbase + ay.oc[sa] + ax.oc[sb];  // (ay.oc: times X2_CP already)
                    const float4 c00 = lds_quad(s_cov + bo * COVSZ + ca), c01 = lds_quad(s_cov + bo * COVSZ + ca + 1);
                    const float4 c10 = lds_quad(s_cov + bo * COVSZ + ca + X2_CP), c11 = lds_quad(s_cov + bo * COVSZ + ca + X2_CP + 1);
                    const float gx = ax.f[sb], gy = ay.f[sa];
                    const float w11 = gx * gy, w01 = gx - w11, w10 = gy - w11, w00 = (1.f - gx) - w10;
                    const float cxx = fmaf(w11, c11.x, fmaf(w10, c10.x, fmaf(w01, c01.x, w00 * c00.x)));
                    const float cxy = fmaf(w11, c11.y, fmaf(w10, c10.y, fmaf(w01, c01.y, w00 * c00.y)));
                    const float cyy = fmaf(w11, c11.w, fmaf(w10, c10.w, fmaf(w01, c01.w, w00 * c00.w)));
                    //@ inverse
                    const float det = fmaf(cxx, cyy, -(cxy * cxy));
                    const float s1 = __builtin_amdgcn_rcpf(det) * X2_KEXP;
                    ixx = s1 * cyy;
                    ixy = (-2.f * s1) * cxy;
                    iyy = s1 * cxx;
                    if (isref_s >= 0) {  // wave-uniform, and kept a real branch by the empty asm statement: if-converted, the reference
                        // frame's rule costs a compare and three v_cndmask (half rate) per sub-pixel of EVERY frame
                        asm volatile("; ref identity");
                        if (!(fabsf(det) > 1e-10f)) {  // linalg.py:53-64: identity (also for NaN, D10)
                            ixx = X2_KEXP;
                            ixy = 0.f;
                            iyy = X2_KEXP;
                        }
                    }
                    // 0 * x is 0 for finite x and NaN for NaN / inf: one NaN test for the three coefficients
                    const float probe = fmaf(0.f, ixx, fmaf(0.f, ixy, 0.f * iyy));
                    finite = probe == probe;
                }
                // the 3 x 3 taps: rows ty + e .. + 2, columns tx + e .. + 2 of the window, as aligned pairs from the
                // copy whose shift makes column tx + e even
                //@ tap_setup
                const int mcol = px + ax.e[sb];  // 0, 1, 2
                const float* __restrict__ rp = ((mcol & 1) ? s_rawB : s_rawA) + bo * RAWSZ + tyrp + (ay.e[sa] * X2_RP + (mcol & 2));  // (scalar part apart)
                const float dx0 = ax.d0[sb], dy0 = ay.d0[sa];
                const float dxs[3] = {dx0 - 1.f, dx0, dx0 + 1.f};
                float sv[2][2], sd[2][2];  // by parity of the tap offset (di + 1, dj + 1)
                // w = exp(-max(0, q) / 2) with Python's max (NaN -> 0, D10).  With finite coefficients q is finite and
                // w = clamp(exp2(z), 0, 1): the clamp is an output modifier of v_exp_f32 (free) and equals the max for
                // z > 0 (non-positive-definite blends at the image border, D11).  Non-finite coefficients (NaN
                // covariances of flat regions, D10; singular hand-made covariances) take the exact per-tap form.
                auto taps = [&](auto exact_c) {
                    //@ taps
                    constexpr bool EXACT = decltype(exact_c)::value;
#pragma unroll
                    for (int di = 0; di < 3; ++di) {
                        const float2 v01 = lds_pair(rp + di * X2_RP), v23 = lds_pair(rp + di * X2_RP + 2);
                        const float c3[3] = {v01.x, v01.y, v23.x};
                        const float dy = di == 1 ? dy0 : dy0 + (float)(di - 1);  // (x + 0.f is an instruction: -0.f + 0.f = +0.f)
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
                //@ tap_setup
#ifdef HHSR_X2_BUDGET
                taps(std::false_type{});
#else
                if (ISO || finite) taps(std::false_type{});
                else taps(std::true_type{});
#endif
                //@ fold
                // tap-offset parity -> absolute raw-coordinate parity (uniform): class (a, b) += r * sv[a ^ by][b ^ bx]
                const int by = (ay.org + py + ay.e[sa]) & 1, bx = (ax.org + px + ax.e[sb]) & 1;
                // (the empty asm statements keep the four arms real branches: if-converted, the permutation costs 16
                // v_cndmask per sub-pixel, twice the FMAs it feeds;
                // and distinct, so that the FMAs are not sunk below the arms leaving 8 permutation moves in each)
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
            }
    };
    //@ loop
    auto frame_n = [&](int n, const float2 fl, float lr, int bo) {
        frame(std::false_type{}, n >= a.n, fl, lr, bo, n);
    };
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
    //@ outside
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


void hhsr_launch_merge_x2(bool iso, bool lmin, dim3 grid, hipStream_t s, const BurstArgs& a, const Geo& g, const Cfa4& c,
                          float* num, float* den) {
    const dim3 block(256);
    if (lmin) {
        if (iso) hipLaunchKernelGGL((k_merge_x2<true, true>), grid, block, 0, s, a, g, c, num, den);
        else hipLaunchKernelGGL((k_merge_x2<false, true>), grid, block, 0, s, a, g, c, num, den);
    } else {
        if (iso) hipLaunchKernelGGL((k_merge_x2<true, false>), grid, block, 0, s, a, g, c, num, den);
        else hipLaunchKernelGGL((k_merge_x2<false, false>), grid, block, 0, s, a, g, c, num, den);
    }
}

/*
 * hhsr.h — C ABI of libhhsr_hip.so: the MI355X (gfx950) kernels of the handheld burst
 * super-resolution hot path.
 *
 * The reference (Jamy-L/Handheld-Multi-Frame-Super-Resolution) has no native layer: its hot path
 * is 25 Numba-CUDA kernels plus torch ops launched from Python.  Each entry point below replaces
 * one of those launch sites (cited as file:line relative to the reference's
 * handheld_super_resolution/ package).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller unless the parameter is documented as
 *    "host"; functions never allocate, free or synchronise (the hhsr_grey_plan_create/destroy pair
 *    excepted): they only enqueue work on `stream` (a hipStream_t passed as void*; NULL = null stream);
 *  - images are row-major float32; `pitch` arguments are in ELEMENTS;
 *  - flow fields are float32 [ny][nx][2] = (dx, dy) per tile, moving(p + flow) ~ ref(p);
 *  - covariances are float32 [H/2][W/2][2][2]; accumulators float32 [sH][sW][3];
 *  - the CFA is 4 bytes {c00, c01, c10, c11} with 0=R, 1=G, 2=B;
 *  - return value: 0 = OK, >0 = hipError_t of the launch, <0 = invalid argument; the message is
 *    available from hhsr_last_error() (thread local).  No C++ exception crosses the boundary.
 *  - no global mutable state: re-entrant across host threads and streams (a grey plan is caller-owned
 *    state).
 */
#ifndef HHSR_H
#define HHSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HHSR_VERSION_MAJOR 0
#define HHSR_VERSION_MINOR 1

#define HHSR_MAX_TAPS 33      /* Gaussian taps: 4*factor+1, factor <= 8 */
#define HHSR_MAX_FRAMES 64    /* frames per hhsr_merge_burst launch */
#define HHSR_MAX_BATCH 8      /* frames per launch of the batched front-end entry points (hhsr_*_batch) */

const char* hhsr_version(void);
const char* hhsr_last_error(void);

/* ---- grey image: mask of the ideal half-band low-pass (utils_image.py:82-100) -----------------
 * The reference zeroes the outer quarter bands of the fftshift-ed full complex spectrum and keeps
 * the real part of the inverse.  hhsr_lowpass_mask_c2c zeroes the same bins of an UN-shifted
 * complex64 spectrum [H][W] in place.  hhsr_lowpass_mask_r2c applies the equivalent Hermitian
 * mask m'(k) = (m(k) + m(-k))/2 in {0, 1/2, 1} to a half spectrum [H][W/2+1] (rfft2 layout), so
 * that irfft2 returns exactly Re(ifft2(mask * fft2(x))). `spec` = interleaved (re, im) float32.
 * stride_y / stride_x: element (complex) strides of the two spectrum dimensions — rocFFT hands torch a
 * transposed half spectrum, which is masked in place as it lies. */
int hhsr_lowpass_mask_c2c(float* spec, int H, int W, int64_t stride_y, int64_t stride_x, void* stream);
int hhsr_lowpass_mask_r2c(float* spec, int H, int W, int64_t stride_y, int64_t stride_x, void* stream);

/* Planned grey transform: real-to-complex rocFFT plan -> Hermitian low-pass mask (normalisation folded
 * in) -> complex-to-real plan; dst = Re(ifft2(mask * fft2(src))), src/dst contiguous [H][W], src intact.
 * hhsr_grey_plan_create is the ONE place the library allocates (the plan and its [H][W/2+1] spectrum
 * buffer, owned by the plan); a plan is mutable state: use it from one stream at a time.
 * Returns 1000 + hipfftResult on a hipFFT error. */
#define HHSR_GREY_PRUNED 1   /* batched row plans + strided column plans on the kept x-bins only (experiment) */
#define HHSR_GREY_FUSED 4    /* three in-LDS kernels (rows, columns + mask, rows) when W is even and W/2, H factor
                                into {2,3,5,7} and fit LDS; otherwise the library plans are used */
#define HHSR_GREY_TPRUNED 2  /* row plans with a transposed spectrum + contiguous column plans on the kept bins */
#define HHSR_GREY_BATCH(n) ((n) << 8)  /* the plan holds n (<= HHSR_MAX_BATCH) spectra: hhsr_grey_lowpass_batch then
                                          transforms n frames with ONE launch per phase (default 1) */
int hhsr_grey_plan_create(int H, int W, int flags, void** plan_out);
int hhsr_grey_lowpass(void* plan, const float* src, float* dst, void* stream);
/* The same transform for n_frames frames (HOST arrays of device pointers): per-frame results are bit-identical to
 * hhsr_grey_lowpass; with the fused kernels the row blocks / kept columns of all frames of a plan-batch share one
 * launch per phase (the per-frame launches are latency-bound: 3 x 19 launches of ~40 us per 12 MP burst). */
int hhsr_grey_lowpass_batch(void* plan, const float* const* srcs, float* const* dsts, int n_frames, void* stream);
int hhsr_grey_plan_destroy(void* plan);

/* ---- pyramid (alignment.py:27-37, 74-82; utils_image.py:360-391) ------------------------------ */
/* dst[y][x] = src[y mod H][x mod W], dst is Hp x Wp (F.pad 'circular', bottom/right). */
int hhsr_pad_circular(const float* src, int H, int W, int src_pitch,
                      float* dst, int Hp, int Wp, int dst_pitch, void* stream);
/* Valid separable Gaussian (rows then columns) + decimation by `factor`:
 * dst is floor((H-2r)/f) x floor((W-2r)/f), r = (ntaps-1)/2; factor 2 or 4 with ntaps = 4*factor+1 (the reference's kernels).  `taps` is a HOST array. */
int hhsr_gauss_decimate(const float* src, int H, int W, int src_pitch,
                        float* dst, int dst_pitch, int factor,
                        const float* taps, int ntaps, void* stream);
/* n_frames levels of the same shape in one launch (HOST arrays of device pointers); per frame bit-identical. */
int hhsr_gauss_decimate_batch(const float* const* srcs, int n_frames, int H, int W, int src_pitch,
                              float* const* dsts, int dst_pitch, int factor,
                              const float* taps, int ntaps, void* stream);

/* ---- Lucas-Kanade precompute (ICA.py:15-76) ---------------------------------------------------
 * gx = I[x+1]-I[x-1], gy likewise (zero border, no 1/2 factor); hess[ty][tx] = sum over the tile of
 * [gx^2, gx gy; gx gy, gy^2] for the floor(H/ts) x floor(W/ts) tile grid. */
int hhsr_grad_hessian(const float* lvl, int H, int W, int pitch, int ts,
                      float* gx, float* gy, float* hess, void* stream);

/* ---- block matching (block_matching.py:20-76, 348-377 and 78-345) -----------------------------
 * L2: argmin over (2r+1)^2 integer shifts of the tile SSD (== the reference's
 * sum(win^2) - 2 corr(ref, win) up to a per-tile constant), window origin
 * tile*ts + round_half_even(flow) - r with clamp-to-edge addressing of the moving level, first
 * minimum in row-major order; the shift is ADDED to the un-rounded flow.
 * L1 (mode 0): the INTENDED semantics of the reference's undefined-behaviour kernels — SAD, zero
 * outside the moving level, flow <- round(flow) + shift.  mode 1: flow <- round_half_even(flow). */
int hhsr_bm_l2(const float* ref, int ref_pitch, const float* mov, int mh, int mw, int mov_pitch,
               float* flow, int ny, int nx, int ts, int r, void* stream);
int hhsr_bm_l1(const float* ref, int ref_pitch, const float* mov, int mh, int mw, int mov_pitch,
               float* flow, int ny, int nx, int ts, int r, int mode, void* stream);

/* ---- ICA (ICA.py:78-482): n_iter Gauss-Newton steps per tile, in place on `flow`.
 * flags bit 0: reproduce the ts=64 row off-by-one of ica_kernel_64 (ICA.py:437-449). */
int hhsr_ica(const float* ref, const float* gx, const float* gy, int ref_pitch, const float* hess,
             const float* mov, int mh, int mw, int mov_pitch,
             float* flow, int ny, int nx, int ts, int n_iter, int flags, void* stream);

/* ---- one pyramid level in one launch: block matching then ICA (alignment.py:125-147), ts in {8, 16, 32}.
 * metric: 0 = L2, 1 = L1 (intended semantics), 2 = L1_ref_effective.  Gradients are taken from the
 * reference level itself (rh x rw) — bit-identical to hhsr_grad_hessian's — `hess` from hhsr_grad_hessian.
 * Same results as hhsr_bm_* followed by hhsr_ica (flags bit 0 is irrelevant below ts = 64).
 * Incoming flow: `flow` itself (coarse_flow NULL, rep >= 0); or the nearest-neighbour upscaling of the coarser
 * level fused in (alignment.py:150-172): mult * coarse_flow[ty/rep][tx/rep] ([cny][cnx][2], zero past it); or
 * zero (coarse_flow NULL, rep < 0: the coarsest level).  `flow` is always the output. */
int hhsr_align_level(const float* ref, int rh, int rw, int ref_pitch, const float* hess,
                     const float* mov, int mh, int mw, int mov_pitch,
                     float* flow, int ny, int nx, int ts, int r, int metric, int n_iter,
                     const float* coarse_flow, int cny, int cnx, int rep, float mult, void* stream);
/* The same level step for n_frames moving frames against ONE reference level in one launch (HOST arrays of device
 * pointers; coarse_flows NULL or one pointer per frame): the coarse levels are 7-27 us launches of a few hundred
 * workgroups each — a chunk of frames fills the GPU where one frame cannot.  Per frame bit-identical. */
int hhsr_align_level_batch(const float* ref, int rh, int rw, int ref_pitch, const float* hess,
                           const float* const* movs, int n_frames, int mh, int mw, int mov_pitch,
                           float* const* flows, int ny, int nx, int ts, int r, int metric, int n_iter,
                           const float* const* coarse_flows, int cny, int cnx, int rep, float mult, void* stream);

/* ---- flow upscaling, nearest mode (alignment.py:150-172): dst[y][x] = mult*src[y/rep][x/rep],
 * zero where y/rep >= sny or x/rep >= snx. */
int hhsr_flow_upscale_nearest(const float* src, int sny, int snx, float* dst, int dny, int dnx,
                              int rep, float mult, void* stream);

/* ---- kernel covariances, Alg. 5 (kernels.py:29-243; utils_image.py:117-170, 346-357;
 * linalg.py:87-185), bayer mode: GAT -> 2x2 mean -> gradients -> structure tensor -> eigen ->
 * (k1, k2) -> covariance per Bayer quad.  law: 0 = hard_threshold, 1 = linear. */
int hhsr_cov_from_raw(const float* raw, int H, int W, int pitch, float* covs,
                      double alpha, double beta, double k_detail, double k_denoise,
                      double D_th, double D_tr, double k_stretch, double k_shrink, int law,
                      void* stream);

/* ---- robustness, Alg. 6-9 (robustness.py) -----------------------------------------------------*/
/* Guide image + 3x3 local mean / variance at guide resolution [3][H/2][W/2]
 * (robustness.py:207-225, 269-294).  wb: HOST double[3].  vars may be NULL (comp frames only need the means). */
int hhsr_rob_stats(const float* raw, int H, int W, int pitch, const uint8_t cfa[4],
                   const double* wb, float* means, float* vars, void* stream);
/* hhsr_rob_stats + hhsr_cov_from_raw in one pass over the raw frame (both work on the same Bayer-quad tile):
 * what super_resolution.py:137-163 computes per frame from the raw image alone.  vars may be NULL. */
int hhsr_frame_stats(const float* raw, int H, int W, int pitch, const uint8_t cfa[4], const double* wb,
                     float* means, float* vars, float* covs, double alpha, double beta, double k_detail,
                     double k_denoise, double D_th, double D_tr, double k_stretch, double k_shrink, int law,
                     void* stream);
/* The same pass for n_frames comp frames (guide means + covariances, no variances) in one launch: HOST arrays of
 * device pointers; per frame bit-identical to hhsr_frame_stats. */
int hhsr_frame_stats_batch(const float* const* raws, int n_frames, int H, int W, int pitch, const uint8_t cfa[4],
                           const double* wb, float* const* means, float* const* covs, double alpha, double beta,
                           double k_detail, double k_denoise, double D_th, double D_tr, double k_stretch,
                           double k_shrink, int law, void* stream);
/* Dodgson-quadratic x2 upsampling of a [3][lh][lw] map to [3][2lh][2lw], optionally warped by the
 * per-tile flow (NULL = reference frame; robustness.py:359-421).  +inf outside. */
int hhsr_rob_upscale(const float* stats, int lh, int lw, const float* flow, int ny, int nx, int ts,
                     float* out, void* stream);
/* Per-tile flow-irregularity map S (robustness.py:570-612): spread of the flow over the 3 x 3 TILE neighbourhood.
 * rows_before / rows_after (0 for a whole field): tile rows that lie in memory before flow[0] / after flow[ny - 1] —
 * `flow` is then a row slice of a larger field (the row slabs of the multi-GPU path, distributed.py) and the
 * neighbourhood reads them, so that the first and last rows of the slice get the weights of the full field. */
int hhsr_rob_s(const float* flow, int ny, int nx, double Mt, float s1, float s2, float* S, int rows_before,
               int rows_after, void* stream);
/* Frame-independent noise-model terms (robustness.py:505-528, once per burst):
 * sigma_sq[p] = sum_c max(ref_vars[c][p], std_curve[round(1000 ref_means[c][p])]^2), float32 [H][W];
 * curve_index[p] (optional, NULL to skip; needs ncurve <= 1024) = the three curve indices
 * round(1000 ref_means[c][p]) packed 10 bits each (c = 0 in the low bits), uint32 [H][W]. */
int hhsr_rob_sigma(const float* ref_means, const float* ref_vars, int H, int W,
                   const double* std_curve, int ncurve, float* sigma_sq, uint32_t* curve_index, void* stream);
/* hhsr_rob_upscale(means) + hhsr_rob_upscale(vars) + hhsr_rob_sigma in one pass over the reference frame's guide
 * statistics [3][lh][lw] (robustness.py:23-76 and 505-528, once per burst): ref_means float32 [3][2lh][2lw],
 * sigma_sq [2lh][2lw], curve_index (optional) — bit-identical to the three calls; the upsampled variances are
 * never written. */
int hhsr_ref_planes(const float* guide_means, const float* guide_vars, int lh, int lw,
                    const double* std_curve, int ncurve, float* ref_means, float* sigma_sq,
                    uint32_t* curve_index, void* stream);
/* Fused warp-upsample of the frame's guide means + colour distance + noise-model shrink + threshold
 * (robustness.py:359-421, 453-461, 505-528, 627-639) -> R float32 [2lh][2lw].
 * ref_sigma_sq / ref_curve_index from hhsr_rob_sigma (index plane NULL = look the curve up per frame on
 * the slower generic kernel); diff_curve: device double[ncurve]. */
int hhsr_rob_frame(const float* comp_means, int lh, int lw, const float* ref_means, const float* ref_sigma_sq,
                   const uint32_t* ref_curve_index, const float* flow, int ny, int nx, int ts, const float* S,
                   const double* diff_curve, int ncurve, double t, float* R, void* stream);
/* hhsr_rob_frame for n_frames frames of one burst (HOST arrays of device pointers): groups of 4 frames share one pass
 * over the reference-frame planes (20 of the 27 bytes per pixel and frame); per frame bit-identical to hhsr_rob_frame.
 * S = NULL: the per-tile weights of hhsr_rob_s are evaluated inside the kernel from (Mt, s1, s2) — one launch less per
 * frame; only with the grouped kernel (ts % 16 == 0, W % 4 == 0, packed curve indices, 16-byte aligned planes; error -3
 * otherwise).  With S given, Mt / s1 / s2 are ignored.  flow_rows_before / flow_rows_after: as in hhsr_rob_s, for the
 * weights evaluated inside the kernel (every flows[n] is a row slice with that many tile rows around it). */
int hhsr_rob_frames(const float* const* comp_means, int n_frames, int lh, int lw, const float* ref_means,
                    const float* ref_sigma_sq, const uint32_t* ref_curve_index, const float* const* flows, int ny,
                    int nx, int ts, const float* const* S, double Mt, float s1, float s2, const double* diff_curve,
                    int ncurve, double t, float* const* R, int flow_rows_before, int flow_rows_after, void* stream);
/* 5x5 clamp-border minimum (robustness.py:670-686).  acc_r != NULL additionally does acc_r += r
 * (the accumulated robustness of super_resolution.py:158-159, fused to save a pass). */
int hhsr_local_min5(const float* R, int H, int W, float* r, float* acc_r, void* stream);
/* Accumulated robustness the way the reference keeps it where something DECIDES on it (super_resolution.py:116-117,
 * 158-159; utils.py:93-120; merge.py:223-228): the float64 sum of the n_frames (<= HHSR_MAX_FRAMES) maps rs[] (HOST array of
 * device pointers, float32 [H][W]) in frame order, in one pass.  load != 0: start from sum64 (bursts longer than one
 * call).  Outputs, each may be NULL (not all): sum64 double [H][W]; mask32 = the sum rounded to float32 (the map the API
 * reports); decisions32 = float32 map a with (a <= mfc) == (sum <= mfc) and (a < mfc) == (sum < mfc) for
 * mfc = max_frame_count — what hhsr_accumulate_ref compares as (double) a (exact for every mfc float32 can hold).
 * flags: HHSR_ROB_SUM_LOAD (= the former `load != 0`) | HHSR_ROB_SUM_MIN5: rs[] are the UN-filtered maps R and the sum is taken
 * over their 5 x 5 clamp-border minima (robustness.py:641-686; = hhsr_local_min5 per frame, then the sum) — for callers whose
 * merge applies the minimum itself (HHSR_MERGE_LOCAL_MIN) and never materialises the filtered maps. */
#define HHSR_ROB_SUM_LOAD 1
#define HHSR_ROB_SUM_MIN5 2
int hhsr_rob_sum(const float* const* rs, int n_frames, int H, int W, int flags, double max_frame_count, double* sum64,
                 float* mask32, float* decisions32, void* stream);

/* ---- monochrome sensors, `mode: grey` (super_resolution.py:106-109, 144-147; kernels.py:83-87; robustness.py:62-66,
 * 145-148, 337-343; merge.py:131-137, 191-194, 349-354, 410): the frame is its own grey image (alignment unchanged), its
 * own one-channel guide image (no white balance) and the kernel covariances are estimated per pixel.
 * hhsr_mono_frame_stats: 3x3 local mean / variance [H][W] and / or covariances [H][W][2][2] in one pass
 *   (means NULL: covariances only; covs NULL: statistics only; vars may be NULL).
 * hhsr_mono_rob_upscale: robustness.py:296-421 on a one-channel map, which keeps its size while the kernel keeps its
 *   hard-coded s = 2 — the top-left quadrant stretched over the frame (+inf outside), reproduced as it is;
 *   flow NULL = the reference frame.
 * hhsr_mono_rob_sigma / hhsr_mono_rob_frame: robustness.py:505-528 / the fused per-frame pass -> R, one channel.
 *   hhsr_mono_rob_frame: with W % 4 == 0, ts % 4 == 0 and 16-byte aligned ref_means / sigma_sq / R the float32
 *   4-pixels-per-thread kernel runs (|dR| <= 1e-4 like the Bayer kernels), otherwise the float64 one-pixel kernel. */
int hhsr_mono_frame_stats(const float* raw, int H, int W, int pitch, float* means, float* vars, float* covs,
                          double alpha, double beta, double k_detail, double k_denoise, double D_th, double D_tr,
                          double k_stretch, double k_shrink, int law, void* stream);
int hhsr_mono_rob_upscale(const float* stats, int H, int W, const float* flow, int ny, int nx, int ts, float* out,
                          void* stream);
int hhsr_mono_rob_sigma(const float* ref_means, const float* ref_vars, int H, int W, const double* std_curve,
                        int ncurve, float* sigma_sq, void* stream);
int hhsr_mono_rob_frame(const float* comp_means, int H, int W, const float* ref_means, const float* sigma_sq,
                        const float* flow, int ny, int nx, int ts, const float* S, const double* diff_curve,
                        int ncurve, double t, float* R, void* stream);

/* kflags of the merge entry points */
#define HHSR_KERNEL_ISO 1   /* merging.kernel == "iso": w = exp(-(dx^2+dy^2)) instead of the steerable kernel */
#define HHSR_WEIGHT_F64 2   /* evaluate covariance interpolation / weights in float64 like the reference's
                               Numba typing (validation mode; default = float32 weights, float64 geometry) */
/* hhsr_merge_burst only — restrict its kernel choice (validation, A/B measurements; default: fastest applicable): */
#define HHSR_MERGE_FORCE_GENERIC 4  /* no LDS staging: one thread per HR pixel, operands from global memory     */
#define HHSR_MERGE_FORCE_TILE 8     /* no x2 kernel: the 16 x 16 HR tile kernel                                  */
#define HHSR_MERGE_FORCE_X2V1 16    /* x2: first-generation kernel (per-pixel geometry) instead of k_merge_x2    */
/* hhsr_accumulate_ref only: */
#define HHSR_REF_DIVIDE 64   /* normalise in the same pass: num = (num (+) ref) / (den (+) ref), den is read and left as it is —
                                accumulate_ref followed by divide (super_resolution.py:187-190) without a second pass over
                                the accumulators; bit-identical to hhsr_accumulate_ref + hhsr_divide                    */
#define HHSR_REF_FAST 128    /* float32 weight chain for the pixels the accumulated-robustness rule leaves alone, outside the
                                border bands — what hhsr_merge_burst does for its reference frame; default: the reference's
                                float64 chain on every pixel                                                            */
/* all three merge entry points: */
#define HHSR_SENSOR_MONO 32  /* `mode: grey`: every sample goes to channel 0; the per-frame entry points and the generic
                                burst kernel leave channels 1, 2 of num / den as they are, the x2 tile kernel of
                                hhsr_merge_burst WRITES them (0, or 0 / 0 = NaN with HHSR_MERGE_DIVIDE — what the reference's
                                three-channel accumulators hold for a monochrome burst); covs is [H][W][2][2] read at the position itself, cfa is ignored (may be NULL);
                                hhsr_merge_burst: scale 2 (ts % 16 == 0, even sizes) runs the LDS-staged x2 tile kernel
                                with a per-pixel covariance window (HHSR_MERGE_LOCAL_MIN available), other scales the
                                generic kernel                                                                     */

/* ---- merge, Alg. 4 / Alg. 11 (merge.py; utils.py:62-120) --------------------------------------
 * hhsr_accumulate: one comp frame, num/den += (merge.py:291-434).
 * hhsr_accumulate_ref: the reference frame (merge.py:83-233); acc_rob = NULL disables the
 * accumulated-robustness widening (rad_max / max_multiplier / max_frame_count ignored). */
int hhsr_accumulate(const float* raw, int H, int W, int pitch, const float* flow, int ny, int nx, int ts,
                    const float* covs, const float* r, const uint8_t cfa[4], double scale, int kflags,
                    float* num, float* den, int sH, int sW, void* stream);
int hhsr_accumulate_ref(const float* raw, int H, int W, int pitch, const float* covs,
                        const uint8_t cfa[4], double scale, int kflags,
                        const float* acc_rob, int rad_max, double max_multiplier, double max_frame_count,
                        float* num, float* den, int sH, int sW, void* stream);
int hhsr_divide(float* num, const float* den, int64_t n, void* stream);   /* num /= den */
int hhsr_add(float* A, const float* B, int64_t n, void* stream);          /* A += B */

/* Fused burst merge: for every HR pixel, sum the contributions of `n_frames` comp frames with the
 * accumulators held in registers (same left-to-right float32 order as n calls of hhsr_accumulate),
 * then optionally add the reference frame and normalise.  HOST arrays of device pointers.
 * Output rows [row0, row0 + nrows) are processed and num / den point at row0 (slabs for the multi-GPU
 * reduce-scatter; row0 = 0, nrows = sH for the whole image).
 * lr_row_offset: 0 for whole frames.  For a SUB-IMAGE (rows [offset, offset + H) of a larger frame: the row slabs of
 * the multi-GPU path) the raw row this image starts at — a multiple of ts and of 2 with an integer offset * scale:
 * sampling positions are then evaluated in full-frame coordinates, so their float64 / float32 roundings (merge.py:113-114
 * keeps idx / scale in float32) and therefore the results do not depend on how the frame was split.
 * acc_r (optional, float32 [H][W], integer scales only) receives sum_n r_n — the accumulated robustness of
 * super_resolution.py:158-159 — at no extra HBM traffic (+= with HHSR_MERGE_LOAD_ACC).
 * Kernel choice (float32 weights): scale 2 and scale 3 with a BAYER cfa (red and blue on one diagonal, green on the other),
 * ts % 16 == 0 and even / exact output sizes run the wave-per-parity-class kernels (three channel accumulators per
 * sub-pixel; scale 3: frames whose window leaves the image are evaluated by the same code with border masks); other
 * 2 x 2 colour layouts and other integer scales the 16 x 16 tile kernels; non-integer scales the per-pixel kernel.
 * flags: */
#define HHSR_MERGE_LOAD_ACC 1   /* start from the existing num/den instead of zero          */
#define HHSR_MERGE_DO_REF 2     /* add the reference frame (ref_raw/ref_covs) after the comps */
#define HHSR_MERGE_DIVIDE 4     /* write num/den into num                                    */
#define HHSR_MERGE_STORE_DEN 8  /* also store den                                            */
#define HHSR_MERGE_LOCAL_MIN 16 /* rs[] hold the thresholded maps R of hhsr_rob_frame; their 5x5 clamp-border minimum
                                   (robustness.py:641-686, hhsr_local_min5) is taken inside the merge.  Only with the
                                   x2 kernels (scale 2, ts % 16 == 0, sH = 2 H, sW = 2 W, row0 % 32 == 0) and the
                                   x3 kernel (scale 3, Bayer cfa, W % 4 == 0, row0 % 48 == 0); float32 weights, no
                                   HHSR_MERGE_FORCE_GENERIC / _TILE; error -3 otherwise.                             */
int hhsr_merge_burst(const float* const* raws, const float* const* flows, const float* const* covs,
                     const float* const* rs, int n_frames, int H, int W, int pitch,
                     int ny, int nx, int ts, const float* ref_raw, const float* ref_covs,
                     const uint8_t cfa[4], double scale, int kflags, int flags,
                     float* num, float* den, float* acc_r, int sH, int sW, int row0, int nrows,
                     int lr_row_offset, void* stream);

/* The fused x2 merge as a CHAIN of launches for bursts whose last frames arrive late (frames crossing PCIe: the early links
 * run while the late frames upload), bit-identical to ONE hhsr_merge_burst over all frames.  Every link gets the frames
 * that have arrived so far, raws[0 .. n_frames), of which the first n_done were merged by the links before it:
 *   first link   flags = HHSR_MERGE_STORE_CLASSES, n_done = 0: frames [0, n_frames) are merged into the kernel's
 *                parity-class accumulators, which are parked in class_acc (hhsr_merge_chain_bytes(H, W) bytes);
 *   middle link  flags = HHSR_MERGE_LOAD_CLASSES | HHSR_MERGE_STORE_CLASSES: restore, add frames [n_done, n_frames), park;
 *   last link    flags = HHSR_MERGE_LOAD_CLASSES | the flags of the single launch (DO_REF, DIVIDE, ...): restore, add the
 *                remaining frames, the reference frame, normalise, write the image (and acc_r).
 * (+ HHSR_MERGE_LOCAL_MIN in every link if the single launch had it.)  Tiles in which a window of ANY frame leaves the
 * image (they run a per-pixel code path: ~2 % at 12 MP) are skipped by the storing links and computed from the first
 * frame on by the last link, so every tile runs exactly the instruction sequence of the single launch.
 * Only with the wave-per-class x2 kernel (scale 2, ts % 16 == 0, sH = 2 H, sW = 2 W, float32 weights, Bayer, whole image);
 * error -3 otherwise. */
#define HHSR_MERGE_STORE_CLASSES 32
#define HHSR_MERGE_LOAD_CLASSES 64
size_t hhsr_merge_chain_bytes(int H, int W);
int hhsr_merge_burst_chain(const float* const* raws, const float* const* flows, const float* const* covs,
                           const float* const* rs, int n_frames, int H, int W, int pitch,
                           int ny, int nx, int ts, const float* ref_raw, const float* ref_covs,
                           const uint8_t cfa[4], double scale, int kflags, int flags,
                           float* num, float* den, float* acc_r, int sH, int sW,
                           float* class_acc, int n_done, void* stream);

/* ---- burst front end (utils_dng.py:149-160) ------------------------------------------------
 * Sensor counts uint16 [n_frames][H][pitch] -> normalised, white-balanced float32 [n_frames][H][W]:
 * v = (float32(count) - black[c]) / (white - black[c]); v *= wb[c] / wb[1], c = cfa[(y&1)*2 + (x&1)], in the
 * reference's float32 arithmetic (bit-identical to its NumPy expression).  black_levels / white_balance: HOST
 * double[3] indexed by colour (R, G, B); raw and out 16-byte aligned. */
int hhsr_normalize_raw_u16(const uint16_t* raw, int n_frames, int H, int W, int pitch, const uint8_t cfa[4],
                           const double* black_levels, double white_level, const double* white_balance,
                           float* out, void* stream);

/* ---- after the path: frame-count denoisers, postprocess, orientation — on the device -------------
 * hhsr_frame_count_denoise: utils_image.py:174-309.  image / out float32 [H][W][3] (the merged image), acc_r float32
 * [ah][aw] (accumulated robustness).  kind 0 = median (strength_max = radius_max <= 7: the reference's 16 x 16 sample
 * buffer overflows beyond that, error -2), kind 1 = gauss (strength_max = sigma_max; window |i|, |j| <= ceil(3 sigma) —
 * the reference's range() of a float does not type under Numba, so the build defines it).  half_index 1 = the
 * reference's index int(round((y - 0.5) / (2 scale))) into acc_r, 0 = the nearest raw pixel, 2 = int(round(y / scale)),
 * the reference's `mode: grey` branch (utils_image.py:203-204, 260-261). */
int hhsr_frame_count_denoise(const float* image, float* out, int H, int W, const float* acc_r, int ah, int aw,
                             double scale, int kind, double strength_max, double max_frame_count, int half_index,
                             void* stream);
/* hhsr_postprocess: raw2rgb.py:206-250 — optional colour matrix cam2rgb (HOST float[9], row major; NULL = off) + clip,
 * optional unsharp mask (skimage.filters.unsharp_mask = scipy.ndimage.gaussian_filter, mode "reflect": DEVICE double
 * taps[2 radius + 1], rows first, float64 accumulation, float32 intermediate in tmp [H][W][3]; result = c + (c - blur)
 * amount), optional devignetting (raw2rgb.py:198-204), clip, optional gamma 1/2.2, clip; the result is stored at its
 * EXIF-oriented position (utils_image.py:12-55; orientation 5..8: out is [W][H][3]).  Tone mapping is out of scope. */
int hhsr_postprocess(const float* image, float* tmp, float* out, int H, int W, const float* cam2rgb, int do_sharpen,
                     double amount, const double* taps, int radius, int do_devignette, int do_gamma, int orientation,
                     void* stream);
/* float32 [H][W] plane (accumulated robustness) to its EXIF-oriented position (utils_image.py:12-55). */
int hhsr_orient_plane(const float* in, float* out, int H, int W, int orientation, void* stream);

/* ---- measurement support ------------------------------------------------------------------------------------------
 * Shader-clock probe: one wave that sleeps for `ticks_100mhz` ticks of the constant 100 MHz counter and stores
 * {shader cycles, 100 MHz ticks} at its start and end into out4 (DEVICE uint64[4]); launched on a side stream around a
 * timed region, (out4[2] - out4[0]) / (out4[3] - out4[1]) x 100 MHz is the clock the kernels next to it ran at
 * (bench.py's "sclk_mhz").  No counterpart in the reference. */
int hhsr_clock_probe(uint64_t* out4, int64_t ticks_100mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HHSR_H */

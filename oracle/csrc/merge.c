/* Oracle, C restatement of the accumulation (Alg. 4 / Alg. 11): reference merge.py:22-434, linalg.py:38-200.
 *
 * TEST INFRASTRUCTURE — never product code.  This file states, pixel by pixel, the SAME float64 / float32 operation
 * sequence as oracle/merge.py (which is the form pinned against the reference's own outputs, tests/golden/merge.npz) so
 * that the randomised sweeps and the full-size comparisons can afford two oracle runs per burst.  It is compiled with
 * -ffp-contract=off and without fast-math: every operation below rounds exactly like the NumPy expression it
 * restates; the one library call, exp(), may differ from NumPy's SIMD exp by an ulp of float64 (tests/test_oracle_kat.py
 * compares the two forms on every scale family: bit-identical or one float32 ulp of an accumulator).
 *
 * Typing follows Numba's: coordinates and weights are float64, the per-pixel val / acc accumulators
 * are float32 and round after every tap (merge.py:429-430).
 */
#include <math.h>
#include <stdint.h>

static inline int64_t i64min(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t i64max(int64_t a, int64_t b) { return a > b ? a : b; }
static inline int64_t clip64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* NumPy's float floor_divide (npy_divmod): what `lr_x // tile_size` evaluates to (merge.py:322-323). */
static double np_floor_divide(double a, double b)
{
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) div -= 1.0;
    }
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
        return fl;
    }
    return copysign(0.0, a / b);
}

/* max(0, z) with Python / Numba semantics: NaN -> 0 (App. A D10; merge.py:424). */
static inline double pymax0(double z) { return z > 0 ? z : 0.0; }

/* Alg. 4, merge.py:291-434 (oracle/merge.py: merge).  covs: float32 [ch, cw, 2, 2]; flow: float32 [ny, nx, 2];
 * num / den: float32 [hr_h, hr_w, 3], updated in place.  rows [row0, row1) of the output only. */
void oracle_merge(const float *comp, int64_t lr_h, int64_t lr_w, const float *flow, int64_t ny, int64_t nx,
                  const float *covs, int64_t ch_, int64_t cw, const float *r, float *num, float *den, int64_t hr_h,
                  int64_t hr_w, const int64_t *cfa, double scale, double ts, int bayer, int iso, int64_t row0, int64_t row1,
                  int threads)
{
    (void)ny; (void)hr_h;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
    for (int64_t hi = row0; hi < row1; ++hi) {
        double lr_y = ((double)hi + 0.5) / scale;
        int64_t py = (int64_t)np_floor_divide(lr_y, ts);
        int64_t i_r = i64min((int64_t)lr_y, lr_h - 1);
        for (int64_t hj = 0; hj < hr_w; ++hj) {
            double lr_x = ((double)hj + 0.5) / scale;
            int64_t px = (int64_t)np_floor_divide(lr_x, ts);
            double flowx = (double)flow[(py * nx + px) * 2 + 0];
            double flowy = (double)flow[(py * nx + px) * 2 + 1];
            int64_t j_r = i64min((int64_t)lr_x, lr_w - 1);
            double local_r = (double)r[i_r * lr_w + j_r];
            double mx = lr_x + flowx, my = lr_y + flowy;
            if (!(mx >= 0 && mx < (double)lr_w && my >= 0 && my < (double)lr_h)) continue; /* merge.py:343-345 */
            double ixx = 0, ixy = 0, iyy = 0;
            if (!iso) {
                double kj = bayer ? mx / 2 - 0.5 : mx - 0.5;
                double ki = bayer ? my / 2 - 0.5 : my - 0.5;
                double fx = kj - trunc(kj), fy = ki - trunc(ki);
                int64_t x0 = i64max((int64_t)trunc(kj), 0), y0 = i64max((int64_t)trunc(ki), 0);
                int64_t x1 = i64min(x0 + 1, cw - 1), y1 = i64min(y0 + 1, ch_ - 1);
                double c3[3];
                static const int ab[3][2] = {{0, 0}, {0, 1}, {1, 1}};
                for (int q = 0; q < 3; ++q) {
                    int o = ab[q][0] * 2 + ab[q][1];
                    float tr = covs[(y0 * cw + x0) * 4 + o], tl = covs[(y0 * cw + x1) * 4 + o];
                    float br = covs[(y1 * cw + x0) * 4 + o], bl = covs[(y1 * cw + x1) * 4 + o];
                    float dt = tl - tr, db = bl - br;                 /* float32 difference, float64 lerp */
                    double top = (double)tr + fx * (double)dt;
                    double bot = (double)br + fx * (double)db;
                    c3[q] = top + fy * (bot - top);
                }
                double det = c3[0] * c3[2] - c3[1] * c3[1];
                double inv_det = 1.0 / det;
                ixx = inv_det * c3[2];
                ixy = -inv_det * c3[1];
                iyy = inv_det * c3[0];
            }
            int64_t cj = (int64_t)trunc(mx), ci = (int64_t)trunc(my);
            double mj = mx - 0.5, mi = my - 0.5;
            float val[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
            for (int di = -1; di <= 1; ++di)
                for (int dj = -1; dj <= 1; ++dj) {
                    int64_t j = cj + dj, i = ci + di;
                    if (!(j >= 0 && j < lr_w && i >= 0 && i < lr_h)) continue;
                    int k = bayer ? (int)cfa[(i % 2) * 2 + (j % 2)] : 0;
                    double c = (double)comp[i * lr_w + j];
                    double dx = (double)j - mj, dy = (double)i - mi;
                    double z;
                    if (iso)
                        z = 2 * (dx * dx + dy * dy);
                    else
                        z = (ixx * dx) * dx + ((2 * ixy) * dx) * dy + (iyy * dy) * dy;
                    z = pymax0(z);
                    double w = exp(-0.5 * z);
                    double wr = w * local_r;
                    val[k] = (float)((double)val[k] + wr * c);
                    acc[k] = (float)((double)acc[k] + wr);
                }
            float *pn = num + (hi * hr_w + hj) * 3, *pd = den + (hi * hr_w + hj) * 3;
            for (int k = 0; k < (bayer ? 3 : 1); ++k) {
                pn[k] = pn[k] + val[k];
                pd[k] = pd[k] + acc[k];
            }
        }
    }
}

/* Alg. 11, merge.py:83-233 (oracle/merge.py: merge_ref).  acc_rob: float64 [H, W] or NULL (denoiser off). */
void oracle_merge_ref(const float *ref, int64_t H, int64_t W, const float *covs, int64_t ch_, int64_t cw, float *num,
                      float *den, int64_t oh, int64_t ow, const int64_t *cfa, double scale, int bayer, int iso,
                      const double *acc_rob, int64_t arh, int64_t arw, double max_frame_count, double max_multiplier,
                      int64_t rad_max, int64_t row0, int64_t row1, int threads)
{
    (void)oh;
    const int denoise = acc_rob != 0;
    const int64_t rmax = denoise ? i64max(rad_max, 1) : 1;
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
    for (int64_t oi = row0; oi < row1; ++oi) {
        float py = (float)((double)oi / scale); /* coarse_ref_sub_pos is a float32 local array */
        for (int64_t oj = 0; oj < ow; ++oj) {
            float px = (float)((double)oj / scale);
            float i00 = 1, i01 = 0, i10 = 0, i11 = 1;
            if (!iso) {
                float gy = bayer ? (float)(((double)py - 0.5) / 2) : py;
                float gx = bayer ? (float)(((double)px - 0.5) / 2) : px;
                int64_t x0 = (int64_t)fmaxf(floorf(gx), 0.0f), y0 = (int64_t)fmaxf(floorf(gy), 0.0f);
                int64_t x1 = i64min(x0 + 1, cw - 1), y1 = i64min(y0 + 1, ch_ - 1);
                double rx = (double)(gx - truncf(gx)), ry = (double)(gy - truncf(gy)); /* modf fraction (signed) */
                float m[4];
                for (int o = 0; o < 4; ++o) {
                    double c00 = (double)covs[(y0 * cw + x0) * 4 + o], c01 = (double)covs[(y0 * cw + x1) * 4 + o];
                    double c10 = (double)covs[(y1 * cw + x0) * 4 + o], c11 = (double)covs[(y1 * cw + x1) * 4 + o];
                    m[o] = (float)(((c00 * (1 - rx)) * (1 - ry) + (c01 * rx) * (1 - ry) + (c10 * (1 - rx)) * ry)
                                   + (c11 * rx) * ry);
                }
                float det = m[0] * m[3] - m[1] * m[2]; /* float32 */
                if (fabsf(det) > 1e-10f) {              /* NaN -> identity; float32 compare like the NumPy form */
                    double det_i = 1 / (double)det;
                    i00 = (float)((double)m[3] * det_i);
                    i01 = (float)(-(double)m[1] * det_i);
                    i10 = (float)(-(double)m[2] * det_i);
                    i11 = (float)((double)m[0] * det_i);
                }
            }
            double power = 1.0;
            int64_t rad = 1;
            int over = 0;
            if (denoise) {
                int64_t ry_i = i64min((int64_t)rint((double)py), arh - 1), rx_i = i64min((int64_t)rint((double)px), arw - 1);
                double lacc = acc_rob[ry_i * arw + rx_i];
                if (lacc <= max_frame_count) { power = max_multiplier; rad = rad_max; }
                over = lacc < max_frame_count;
            }
            int64_t cx = (int64_t)rintf(px), cy = (int64_t)rintf(py);
            float val[3] = {0, 0, 0}, acc[3] = {0, 0, 0};
            for (int64_t i = -rmax; i <= rmax; ++i)
                for (int64_t j = -rmax; j <= rmax; ++j) {
                    int64_t pj = cx + j, pi = cy + i;
                    int64_t ai = i < 0 ? -i : i, aj = j < 0 ? -j : j;
                    if (!(ai <= rad && aj <= rad && pj >= 0 && pj < W && pi >= 0 && pi < H)) continue;
                    int k = bayer ? (int)cfa[(pi % 2) * 2 + (pj % 2)] : 0;
                    double c = (double)ref[pi * W + pj];
                    double dx = (double)pj - (double)px, dy = (double)pi - (double)py;
                    double y;
                    if (iso)
                        y = pymax0(2 * (dx * dx + dy * dy));
                    else {
                        float s01 = i01 + i10; /* float32 sum, then float64 */
                        y = pymax0(((double)i00 * dx) * dx + (dx * dy) * (double)s01 + ((double)i11 * dy) * dy);
                    }
                    y = y / power;
                    double w = exp(-0.5 * y);
                    val[k] = (float)((double)val[k] + c * w);
                    acc[k] = (float)((double)acc[k] + w);
                }
            float *pn = num + (oi * ow + oj) * 3, *pd = den + (oi * ow + oj) * 3;
            for (int k = 0; k < (bayer ? 3 : 1); ++k) {
                pn[k] = over ? val[k] : pn[k] + val[k];
                pd[k] = over ? acc[k] : pd[k] + acc[k];
            }
        }
    }
}

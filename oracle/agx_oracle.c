/*
 * agx_oracle.c -- CPU restatement of the aprilgrid 0.8.0 detection path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load this library, and
 * only as the checker / the timed CPU baseline.  The shipped path (aprilgrid-rs_amd/)
 * never links, imports or calls anything in oracle/.
 *
 * Each function cites the reference lines (under /root/reference) whose arithmetic it
 * restates.  All arithmetic is IEEE binary32, one rounding per operation, no fused
 * multiply-add and no reassociation (build with -ffp-contract=off, no -ffast-math):
 * that is what rustc emits for the reference.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - pinned by the reference's own fixtures: the 7 tag-count assertions of
 *     tests/test_detector.rs:26-32 (run end to end through this file), the
 *     hessian_response / pixel_bfs / find_xy / theta_distance / angle / cross / dot /
 *     is_valid_quad / tag_affine known answers of the reference's unit tests.
 *   - PARITY UNPINNED: blur values, saddle coordinates, tag ids and tag corner
 *     coordinates are asserted by no reference test, and the third-party arithmetic
 *     this file restates from the crates' published behaviour (image 0.25.9 luma
 *     conversions, faer 0.23.2 QR / LU, kdtree 0.8.0 k-NN order) could not be executed
 *     here (no Rust toolchain, crates not vendored).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_FMT_L8 0
#define ORC_FMT_L16 1
#define ORC_FMT_RGB8 2
#define ORC_FMT_LF32 3 /* the caller's own DynamicImage::to_luma32f plane (any other variant), taken as is */

typedef struct {
    float x, y, k, theta, phi;
} orc_saddle; /* src/saddle.rs:3-9 : p.0, p.1, k, theta, phi */

typedef struct {
    uint32_t id;
    float xy[8]; /* 4 corners (x,y) in the order detect() returns them */
} orc_tag;

typedef struct {
    float tag_spacing_ratio; /* src/detector.rs:26 (dead in the reference) */
    float min_saddle_angle;  /* :27 */
    float max_saddle_angle;  /* :28 */
    int max_num_of_boards;   /* :29 */
} orc_params;

static const float ORC_PI = 3.14159274101257324219f; /* std::f32::consts::PI */

/* ------------------------------------------------------------------------------------
 * Front end: image 0.25.9 DynamicImage::to_luma32f / to_luma8 (call sites
 * src/detector.rs:409 and :507).  Restated from the crate's published behaviour
 * (SURVEY.md Appendix B): the crate source is not on this machine.
 * ---------------------------------------------------------------------------------- */
static inline uint8_t rgb_to_luma_u8(uint8_t r, uint8_t g, uint8_t b)
{
    /* image: SRGB_LUMA = [2126, 7152, 722], SRGB_LUMA_DIV = 10000, u32 arithmetic */
    uint32_t l = 2126u * r + 7152u * g + 722u * b;
    return (uint8_t)(l / 10000u);
}

int orc_luma_f32(const void *pixels, int w, int h, long stride_bytes, int fmt, float *out)
{
    for (int y = 0; y < h; ++y) {
        const uint8_t *row = (const uint8_t *)pixels + (size_t)y * (size_t)stride_bytes;
        float *o = out + (size_t)y * w;
        if (fmt == ORC_FMT_L8) {
            for (int x = 0; x < w; ++x) o[x] = (float)row[x] / 255.0f;
        } else if (fmt == ORC_FMT_L16) {
            const uint16_t *r16 = (const uint16_t *)row;
            for (int x = 0; x < w; ++x) o[x] = (float)r16[x] / 65535.0f;
        } else if (fmt == ORC_FMT_RGB8) {
            for (int x = 0; x < w; ++x)
                o[x] = (float)rgb_to_luma_u8(row[3 * x], row[3 * x + 1], row[3 * x + 2]) / 255.0f;
        } else if (fmt == ORC_FMT_LF32) {
            memcpy(o, row, (size_t)w * sizeof(float));
        } else {
            return -1;
        }
    }
    return 0;
}

int orc_luma_u8(const void *pixels, int w, int h, long stride_bytes, int fmt, uint8_t *out)
{
    for (int y = 0; y < h; ++y) {
        const uint8_t *row = (const uint8_t *)pixels + (size_t)y * (size_t)stride_bytes;
        uint8_t *o = out + (size_t)y * w;
        if (fmt == ORC_FMT_L8) {
            memcpy(o, row, (size_t)w);
        } else if (fmt == ORC_FMT_L16) {
            const uint16_t *r16 = (const uint16_t *)row;
            for (int x = 0; x < w; ++x) o[x] = (uint8_t)(((uint32_t)r16[x] + 128u) / 257u);
        } else if (fmt == ORC_FMT_RGB8) {
            for (int x = 0; x < w; ++x)
                o[x] = rgb_to_luma_u8(row[3 * x], row[3 * x + 1], row[3 * x + 2]);
        } else {
            return -1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * gaussian_blur_f32 -- src/image_util.rs:110-206
 * ---------------------------------------------------------------------------------- */
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* kernel weights, src/image_util.rs:111-124.  Returns the radius; weights has 2r+1 taps. */
int orc_blur_weights(float sigma, float *weights, int cap)
{
    int radius = (int)ceilf(sigma * 2.0f);
    int size = radius * 2 + 1;
    if (size > cap) return -1;
    float two_sigma_sq = 2.0f * sigma * sigma;
    float sum = 0.0f;
    for (int i = 0; i < size; ++i) {
        float x = (float)(i - radius);
        float v = expf(-(x * x) / two_sigma_sq);
        weights[i] = v;
        sum += v;
    }
    for (int i = 0; i < size; ++i) weights[i] /= sum;
    return radius;
}

void orc_gaussian_blur_f32(const float *img, int w, int h, float sigma, float *out)
{
    float kernel[64];
    int radius = orc_blur_weights(sigma, kernel, 64);
    int size = radius * 2 + 1;
    float *temp = (float *)calloc((size_t)w * h, sizeof(float));
    memset(out, 0, (size_t)w * h * sizeof(float));

    /* horizontal pass, :137-185, in the reference's three regions: left border (:144-153) and right border
     * (:170-184, skipping columns the left loop has already written when w < 2r, :173-175) clamp every tap;
     * the centre (:156-167, only when w > 2r) never clamps and is written to vectorise, as the reference's is --
     * it matters for the cpu_baseline timing, not for the values: all three evaluate
     * val = 0; val += img[clamp(x+i-r)] * k[i] for i ascending. */
    const int left_end = radius < w ? radius : w;
    const int right_start = w - radius > 0 ? w - radius : 0;
    for (int y = 0; y < h; ++y) {
        const float *restrict row = img + (size_t)y * w;
        float *restrict trow = temp + (size_t)y * w;
        for (int x = 0; x < left_end; ++x) {
            float val = 0.0f;
            for (int i = 0; i < size; ++i) val += row[clampi(x + i - radius, 0, w - 1)] * kernel[i];
            trow[x] = val;
        }
        if (w > 2 * radius) {
            const int n_centre = w - 2 * radius;
            if (size == 7) { /* sigma 1.5, the only value the detector uses: the taps unrolled */
                const float k0 = kernel[0], k1 = kernel[1], k2 = kernel[2], k3 = kernel[3], k4 = kernel[4], k5 = kernel[5], k6 = kernel[6];
                for (int i = 0; i < n_centre; ++i) {
                    float val = 0.0f;
                    val += row[i] * k0;
                    val += row[i + 1] * k1;
                    val += row[i + 2] * k2;
                    val += row[i + 3] * k3;
                    val += row[i + 4] * k4;
                    val += row[i + 5] * k5;
                    val += row[i + 6] * k6;
                    trow[radius + i] = val;
                }
            } else {
                for (int i = 0; i < n_centre; ++i) {
                    float val = 0.0f;
                    for (int k = 0; k < size; ++k) val += row[i + k] * kernel[k];
                    trow[radius + i] = val;
                }
            }
        }
        for (int x = right_start; x < w; ++x) {
            if (x < radius) continue;
            float val = 0.0f;
            for (int i = 0; i < size; ++i) val += row[clampi(x + i - radius, 0, w - 1)] * kernel[i];
            trow[x] = val;
        }
    }
    /* vertical pass, :187-203: out row accumulates the 2r+1 clamped temp rows in tap order */
    for (int y = 0; y < h; ++y) {
        float *orow = out + (size_t)y * w;
        for (int i = 0; i < size; ++i) {
            int ky = clampi(y + i - radius, 0, h - 1);
            const float *trow = temp + (size_t)ky * w;
            float kw = kernel[i];
            for (int x = 0; x < w; ++x) orow[x] += trow[x] * kw;
        }
    }
    free(temp);
}

/* ------------------------------------------------------------------------------------
 * hessian_response -- src/image_util.rs:72-109
 * ---------------------------------------------------------------------------------- */
void orc_hessian_response(const float *img, int w, int h, float *out)
{
    memset(out, 0, (size_t)w * h * sizeof(float));
    for (int r = 1; r < h - 1; ++r) {
        const float *p = img + (size_t)(r - 1) * w;
        const float *c = img + (size_t)r * w;
        const float *n = img + (size_t)(r + 1) * w;
        for (int x = 1; x < w - 1; ++x) {
            float v11 = p[x - 1], v12 = p[x], v13 = p[x + 1];
            float v21 = c[x - 1], v22 = c[x], v23 = c[x + 1];
            float v31 = n[x - 1], v32 = n[x], v33 = n[x + 1];
            float lxx = v21 - (v22 * 2.0f) + v23;           /* :100 */
            float lyy = v12 - (v22 * 2.0f) + v32;           /* :101 */
            float lxy = (v13 - v11 + v31 - v33) * 0.25f;    /* :102 */
            out[(size_t)r * w + x] = lxx * lyy - lxy * lxy; /* :104 */
        }
    }
}

/* min fold, src/detector.rs:414-417 */
float orc_min_response(const float *resp, size_t n)
{
    /* fold(f32::MAX, |acc, &e| acc.min(e)) with f32::min inline, as rustc emits it -- a call to libm's fminf per pixel
     * cost 1.6 of the chain's 7 ms and is not what the reference executes.  f32::min returns the other argument when one
     * is NaN; acc starts at f32::MAX and therefore never becomes NaN, so `e < acc ? e : acc` (false for a NaN e: acc
     * stays) is that function exactly -- one minss.  Still the reference's sequential left fold: no reassociation. */
    float acc = 3.40282346638528859812e+38f; /* f32::MAX */
    for (size_t i = 0; i < n; ++i) {
        const float e = resp[i];
        acc = e < acc ? e : acc;
    }
    return acc;
}

/* ------------------------------------------------------------------------------------
 * pixel_bfs -- src/image_util.rs:208-236 ; init_saddle_clusters -- src/detector.rs:171-187
 * ---------------------------------------------------------------------------------- */
typedef struct {
    uint32_t *xy; /* pairs */
    size_t n, cap;
} u32pairs;

static void pairs_push(u32pairs *v, uint32_t x, uint32_t y)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        v->xy = (uint32_t *)realloc(v->xy, v->cap * 2 * sizeof(uint32_t));
    }
    v->xy[2 * v->n] = x;
    v->xy[2 * v->n + 1] = y;
    v->n++;
}

/* appends the member pixels (in visit order) to `cluster`; mat is modified in place */
static void pixel_bfs(float *mat, uint32_t w, uint32_t h, u32pairs *cluster, u32pairs *stack,
                      uint32_t x, uint32_t y, float threshold)
{
    stack->n = 0;
    pairs_push(stack, x, y);
    while (stack->n) {
        stack->n--;
        uint32_t cx = stack->xy[2 * stack->n], cy = stack->xy[2 * stack->n + 1];
        if (cx >= w || cy >= h) continue; /* :217 */
        float v = mat[(size_t)cy * w + cx];
        if (v < threshold) {
            pairs_push(cluster, cx, cy);
            mat[(size_t)cy * w + cx] = 3.40282346638528859812e+38f; /* :224 */
            if (cx > 0) pairs_push(stack, cx - 1, cy);
            pairs_push(stack, cx + 1, cy);
            if (cy > 0) pairs_push(stack, cx, cy - 1);
            pairs_push(stack, cx, cy + 1);
        }
    }
}

/* Public wrapper of pixel_bfs for the known-answer test of src/image_util.rs:296-316.
 * Returns the cluster size; out_xy receives up to cap (x,y) pairs. */
int orc_pixel_bfs(float *mat, int w, int h, int x, int y, float threshold, uint32_t *out_xy,
                  int cap)
{
    u32pairs cluster = {0}, stack = {0};
    pixel_bfs(mat, (uint32_t)w, (uint32_t)h, &cluster, &stack, (uint32_t)x, (uint32_t)y,
              threshold);
    int n = (int)cluster.n;
    for (int i = 0; i < n && i < cap; ++i) {
        out_xy[2 * i] = cluster.xy[2 * i];
        out_xy[2 * i + 1] = cluster.xy[2 * i + 1];
    }
    free(cluster.xy);
    free(stack.xy);
    return n;
}

/* init_saddle_clusters + centroid map (src/detector.rs:171-187, :421-429).
 * h_mat is consumed (mutated).  Writes cluster centres (x,y) and, optionally, the first
 * pixel's linear index and the size of every cluster.  Returns the number of clusters
 * found (may exceed cap; only the first cap are written). */
int orc_cluster_centers(float *h_mat, int w, int h, float threshold, float *centers_xy,
                        uint32_t *first_index, uint32_t *sizes, int cap)
{
    u32pairs cluster = {0}, stack = {0};
    int n_clusters = 0;
    for (int r = 1; r < h - 1; ++r) {
        for (int c = 1; c < w - 1; ++c) {
            float v = h_mat[(size_t)r * w + c];
            if (v < threshold) {
                cluster.n = 0;
                pixel_bfs(h_mat, (uint32_t)w, (uint32_t)h, &cluster, &stack, (uint32_t)c,
                          (uint32_t)r, threshold);
                if (cluster.n) {
                    if (n_clusters < cap) {
                        /* :424-427 f32 running sums in visit order */
                        float sx = 0.0f, sy = 0.0f;
                        for (size_t i = 0; i < cluster.n; ++i) {
                            sx = sx + (float)cluster.xy[2 * i];
                            sy = sy + (float)cluster.xy[2 * i + 1];
                        }
                        centers_xy[2 * n_clusters] = sx / (float)cluster.n;
                        centers_xy[2 * n_clusters + 1] = sy / (float)cluster.n;
                        if (first_index) first_index[n_clusters] = (uint32_t)r * (uint32_t)w + (uint32_t)c;
                        if (sizes) sizes[n_clusters] = (uint32_t)cluster.n;
                    }
                    n_clusters++;
                }
            }
        }
    }
    free(cluster.xy);
    free(stack.xy);
    return n_clusters;
}

/* ------------------------------------------------------------------------------------
 * math_util.rs
 * ---------------------------------------------------------------------------------- */

/* find_xy -- src/math_util.rs:5-12.  The reference calls faer's partial-pivot LU on a
 * 2x2; restated here as textbook LU with row pivoting (pivot = larger |a| in column 0,
 * first row on a tie), forward then back substitution with true divisions.  faer's
 * exact operation order (it may scale by a reciprocal) is unpinned. */
void orc_find_xy(float a0, float b0, float c0, float a1, float b1, float c1, float *x, float *y)
{
    float r0 = -c0, r1 = -c1;
    float pa, pb, pr, qa, qb, qr;
    if (fabsf(a1) > fabsf(a0)) {
        pa = a1; pb = b1; pr = r1; qa = a0; qb = b0; qr = r0;
    } else {
        pa = a0; pb = b0; pr = r0; qa = a1; qb = b1; qr = r1;
    }
    float l = qa / pa;
    float u22 = qb - l * pb;
    float y2 = qr - l * pr;
    float yy = y2 / u22;
    float xx = (pr - pb * yy) / pa;
    *x = xx;
    *y = yy;
}

/* theta_distance_degree -- src/math_util.rs:15-23 */
float orc_theta_distance_degree(float t0, float t1)
{
    float d = t0 - t1 + 90.0f;
    if (d < 0.0f) d += 180.0f;
    else if (d > 180.0f) d -= 180.0f;
    return d > 90.0f ? d - 90.0f : 90.0f - d;
}
/* cross / dot / angle_degree -- src/math_util.rs:24-33 */
float orc_cross(float v0x, float v0y, float v1x, float v1y) { return v0x * v1y - v0y * v1x; }
float orc_dot(float v0x, float v0y, float v1x, float v1y) { return v0x * v1x + v0y * v1y; }
float orc_angle_degree(float v0x, float v0y, float v1x, float v1y)
{
    return atan2f(v1y * v0x - v1x * v0y, v0x * v1x + v0y * v1y) * 180.0f / ORC_PI;
}

/* ------------------------------------------------------------------------------------
 * rochade_refine -- src/detector.rs:194-361
 * ---------------------------------------------------------------------------------- */

/* p_mat (25x6, stored [i*6+j]) = pseudo-inverse of the quadratic design matrix,
 * src/detector.rs:208-237.  The reference gets it from faer's f32 Householder QR (not
 * bit-reproducible without faer); here: normal equations in binary64 solved by
 * Gauss-Jordan with pivoting, then one rounding to binary32 -- i.e. the correctly
 * rounded exact pseudo-inverse.  Expected deviation from faer: last few ulp. */
void orc_refine_pmat(int half, float *p_mat /* n*6 */, float *flat_k /* n */)
{
    int ks = half * 2 + 1, n = ks * ks;
    double *A = (double *)malloc(sizeof(double) * n * 6);
    int count = 0;
    for (int r = 0; r < ks; ++r)
        for (int c = 0; c < ks; ++c) {
            double x = (double)c - half, y = (double)r - half;
            A[count * 6 + 0] = x * x;
            A[count * 6 + 1] = x * y;
            A[count * 6 + 2] = y * y;
            A[count * 6 + 3] = x;
            A[count * 6 + 4] = y;
            A[count * 6 + 5] = 1.0;
            count++;
        }
    double M[6][12];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += A[k * 6 + i] * A[k * 6 + j];
            M[i][j] = s;
            M[i][6 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 6; ++col) {
        int piv = col;
        for (int r = col + 1; r < 6; ++r)
            if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (piv != col)
            for (int j = 0; j < 12; ++j) {
                double t = M[col][j];
                M[col][j] = M[piv][j];
                M[piv][j] = t;
            }
        double d = M[col][col];
        for (int j = 0; j < 12; ++j) M[col][j] /= d;
        for (int r = 0; r < 6; ++r)
            if (r != col) {
                double f = M[r][col];
                for (int j = 0; j < 12; ++j) M[r][j] -= f * M[col][j];
            }
    }
    /* pinv = (A^T A)^-1 A^T  -> p_mat[i][j] = pinv[j][i] */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += M[j][6 + k] * A[i * 6 + k];
            p_mat[i * 6 + j] = (float)s;
        }
    free(A);
    /* cone kernel, :240-254 */
    float gamma = (float)half;
    float s = 0.0f;
    for (int i = 0; i < ks; ++i)
        for (int j = 0; j < ks; ++j) {
            float v = fmaxf(0.0f, gamma + 1.0f -
                                      sqrtf((gamma - (float)i) * (gamma - (float)i) +
                                            (gamma - (float)j) * (gamma - (float)j)));
            flat_k[i * ks + j] = v;
        }
    for (int i = 0; i < n; ++i) s += flat_k[i]; /* iter().sum::<f32>() : left fold from 0.0 */
    for (int i = 0; i < n; ++i) flat_k[i] = flat_k[i] / s;
}

/* Returns the number of refined saddles written to out (at most n_centers). */
int orc_rochade_refine(const float *image, int width, int height, const float *centers_xy,
                       int n_centers, int half, orc_saddle *out)
{
    const float PIXEL_MOVE_THRESHOLD = 1.0f;
    int ks = half * 2 + 1, np = ks * ks;
    float *p_mat = (float *)malloc(sizeof(float) * np * 6);
    float *flat_k = (float *)malloc(sizeof(float) * np);
    float *smooth = (float *)malloc(sizeof(float) * np);
    orc_refine_pmat(half, p_mat, flat_k);
    int half2 = half * 2;
    int n_out = 0;
    for (int ci = 0; ci < n_centers; ++ci) {
        float initial_x = centers_xy[2 * ci], initial_y = centers_xy[2 * ci + 1];
        int round_x = (int)roundf(initial_x);
        int round_y = (int)roundf(initial_y);
        if (round_y - half2 < 0 || round_y + half2 >= height || round_x - half2 < 0 ||
            round_x + half2 >= width) /* :268-274 */
            continue;
        size_t start_x = (size_t)(round_x - half2), start_y = (size_t)(round_y - half2);
        /* :283-317 cone-filtered (2h+1)^2 patch; one running sum per output, taps in
         * (pr, pc) order */
        for (int r = 0; r < ks; ++r)
            for (int c = 0; c < ks; ++c) {
                float conv_p = 0.0f;
                int k_idx = 0;
                for (int pr = 0; pr < ks; ++pr) {
                    const float *rp = image + (start_y + r + pr) * (size_t)width + start_x + c;
                    for (int pc = 0; pc < ks; ++pc) {
                        conv_p += rp[pc] * flat_k[k_idx];
                        k_idx++;
                    }
                }
                smooth[r * ks + c] = conv_p;
            }
        /* :321-328 */
        float params[6];
        for (int j = 0; j < 6; ++j) {
            float sum = 0.0f;
            for (int i = 0; i < np; ++i) sum += p_mat[i * 6 + j] * smooth[i];
            params[j] = sum;
        }
        float a1 = params[0], a2 = params[1], a3 = params[2], a4 = params[3], a5 = params[4];
        float fxx = 2.0f * a1, fyy = 2.0f * a3, fxy = a2;
        float d = fxx * fyy - fxy * fxy;
        if (d < 0.0f) {
            float x0, y0;
            orc_find_xy(2.0f * a1, a2, a4, a2, 2.0f * a3, a5, &x0, &y0);
            if (fabsf(x0) <= PIXEL_MOVE_THRESHOLD && fabsf(y0) <= PIXEL_MOVE_THRESHOLD) {
                float c5 = (a1 + a3) / 2.0f;
                float c4 = (a1 - a3) / 2.0f;
                float c3 = a2 / 2.0f;
                float k = sqrtf(c4 * c4 + c3 * c3);
                if (fabsf(c5) < k) {
                    float phi = acosf(-c5 / k) / 2.0f / ORC_PI * 180.0f;
                    float theta = atan2f(c3, c4) / 2.0f / ORC_PI * 180.0f;
                    out[n_out].x = roundf(initial_x) + x0;
                    out[n_out].y = roundf(initial_y) + y0;
                    out[n_out].k = k;
                    out[n_out].theta = theta;
                    out[n_out].phi = phi;
                    n_out++;
                }
            }
        }
    }
    free(p_mat);
    free(flat_k);
    free(smooth);
    return n_out;
}

/* ------------------------------------------------------------------------------------
 * TagDetector::refined_saddle_points -- src/detector.rs:408-446
 *
 * Optional debug outputs (any may be NULL): blur_out / resp_out (w*h floats each, resp
 * BEFORE the flood fill overwrites it), min_out, n_clusters_out, centers_out (cap_c
 * pairs), first_index_out / sizes_out (cap_c), n_refined_out + refined_out (unfiltered
 * rochade_refine output, cap_c records).
 * Returns the number of saddles after the k / phi filter (may exceed cap; only cap
 * written) or a negative error.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    float *blur;
    float *resp;
    float *min_resp;
    int *n_clusters;
    float *centers;
    uint32_t *first_index;
    uint32_t *sizes;
    int cap_clusters;
    int *n_refined;
    orc_saddle *refined;
} orc_debug;

int orc_refined_saddle_points(const void *pixels, int w, int h, long stride_bytes, int fmt,
                              const orc_params *prm, orc_saddle *out, int cap, orc_debug *dbg)
{
    if (w < 2 || h < 2) return -2; /* reference underflows height()-1 here (panic) */
    size_t n = (size_t)w * h;
    float *luma = (float *)malloc(n * sizeof(float));
    float *blur = (float *)malloc(n * sizeof(float));
    float *resp = (float *)malloc(n * sizeof(float));
    if (orc_luma_f32(pixels, w, h, stride_bytes, fmt, luma)) {
        free(luma); free(blur); free(resp);
        return -1;
    }
    orc_gaussian_blur_f32(luma, w, h, 1.5f, blur);
    orc_hessian_response(blur, w, h, resp);
    float min_response = orc_min_response(resp, n);
    float thr = min_response * 0.05f;
    if (dbg && dbg->blur) memcpy(dbg->blur, blur, n * sizeof(float));
    if (dbg && dbg->resp) memcpy(dbg->resp, resp, n * sizeof(float));
    if (dbg && dbg->min_resp) *dbg->min_resp = min_response;

    /* cluster count is not known in advance: two passes would mutate resp twice, so size
     * generously (one cluster needs at least one pixel; clusters are separated) */
    int cap_c = (int)(n / 2 + 16);
    float *centers = (float *)malloc(sizeof(float) * 2 * (size_t)cap_c);
    uint32_t *first_index = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cap_c);
    uint32_t *sizes = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cap_c);
    int nc = orc_cluster_centers(resp, w, h, thr, centers, first_index, sizes, cap_c);
    if (dbg && dbg->n_clusters) *dbg->n_clusters = nc;
    if (dbg && dbg->centers)
        for (int i = 0; i < nc && i < dbg->cap_clusters; ++i) {
            dbg->centers[2 * i] = centers[2 * i];
            dbg->centers[2 * i + 1] = centers[2 * i + 1];
            if (dbg->first_index) dbg->first_index[i] = first_index[i];
            if (dbg->sizes) dbg->sizes[i] = sizes[i];
        }
    orc_saddle *sp = (orc_saddle *)malloc(sizeof(orc_saddle) * (size_t)(nc > 0 ? nc : 1));
    int ns = orc_rochade_refine(blur, w, h, centers, nc, 2, sp);
    if (dbg && dbg->n_refined) *dbg->n_refined = ns;
    if (dbg && dbg->refined)
        for (int i = 0; i < ns && i < dbg->cap_clusters; ++i) dbg->refined[i] = sp[i];

    int n_out = 0;
    if (ns > 0) {
        float max_k = -3.40282346638528859812e+38f; /* f32::MIN */
        for (int i = 0; i < ns; ++i) max_k = fmaxf(max_k, sp[i].k);
        float s_max_k = max_k / 10.0f;
        for (int i = 0; i < ns; ++i) {
            if (sp[i].k >= s_max_k && sp[i].phi >= prm->min_saddle_angle &&
                sp[i].phi <= prm->max_saddle_angle) {
                if (n_out < cap) out[n_out] = sp[i];
                n_out++;
            }
        }
    }
    free(sp); free(centers); free(first_index); free(sizes);
    free(luma); free(blur); free(resp);
    return n_out;
}

/* ====================================================================================
 * Host tail: board search and tag decode.
 * ==================================================================================== */

/* is_valid_quad -- src/saddle.rs:17-67 */
int orc_is_valid_quad(const orc_saddle *s0, const orc_saddle *d0, const orc_saddle *s1,
                      const orc_saddle *d1)
{
    if (orc_theta_distance_degree(d0->theta, d1->theta) > 5.0f) return 0;
    float v01x = d0->x - s0->x, v01y = d0->y - s0->y;
    float v03x = d1->x - s0->x, v03y = d1->y - s0->y;
    float v02x = s1->x - s0->x, v02y = s1->y - s0->y;
    float s0_theta = s0->theta / 180.0f * ORC_PI;
    float vtx = cosf(s0_theta), vty = sinf(s0_theta);
    float angle = fabsf(orc_angle_degree(v02x, v02y, vtx, vty));
    if (!(angle >= 60.0f && angle <= 120.0f)) return 0;
    float c0 = orc_cross(v01x, v01y, v02x, v02y);
    float c1 = orc_cross(v02x, v02y, v03x, v03y);
    if (c0 * c1 < 0.0f) return 0;
    float v12x = s1->x - d0->x, v12y = s1->y - d0->y;
    float v23x = d1->x - s1->x, v23y = d1->y - s1->y;
    float c01 = orc_cross(v01x, v01y, v12x, v12y);
    float c12 = orc_cross(v12x, v12y, v23x, v23y);
    if (c01 * c12 < 0.0f) return 0;
    float v30x = s0->x - d1->x, v30y = s0->y - d1->y;
    float a0 = orc_angle_degree(v01x, v01y, v12x, v12y);
    float a1 = orc_angle_degree(v12x, v12y, v23x, v23y);
    float a2 = orc_angle_degree(v23x, v23y, v30x, v30y);
    float a3 = orc_angle_degree(v30x, v30y, v01x, v01y);
    if (fabsf(a0 - a2) > 10.0f || fabsf(a1 - a3) > 10.0f) return 0;
    if (orc_dot(v01x, v01y, v02x, v02y) < 0.0f || orc_dot(v03x, v03y, v02x, v02y) < 0.0f) return 0;
    return 1;
}

/* kdtree 0.8.0 KdTree::nearest(point, n, squared_euclidean): the n nearest entries in
 * ascending distance order (restated as an exhaustive search; exact-distance ties, whose
 * order the crate leaves to its heap, are broken by ascending index here).
 * squared_euclidean folds (a-b)^2 from 0.0 in f32. */
typedef struct {
    float d;
    int idx;
} nn_t;

static int nn_cmp(const void *a, const void *b)
{
    const nn_t *p = (const nn_t *)a, *q = (const nn_t *)b;
    if (p->d < q->d) return -1;
    if (p->d > q->d) return 1;
    return p->idx - q->idx;
}

static int knn(const orc_saddle *pts, int n, float qx, float qy, int k, nn_t *out, nn_t *scratch)
{
    for (int i = 0; i < n; ++i) {
        float dx = qx - pts[i].x, dy = qy - pts[i].y;
        scratch[i].d = (0.0f + dx * dx) + dy * dy;
        scratch[i].idx = i;
    }
    qsort(scratch, (size_t)n, sizeof(nn_t), nn_cmp);
    int m = k < n ? k : n;
    memcpy(out, scratch, sizeof(nn_t) * (size_t)m);
    return m;
}

/* ---- Board: src/board.rs ---------------------------------------------------------- */
typedef struct {
    int x, y;
    int has;    /* Some / None */
    int q[4];
} board_cell;

typedef struct {
    const orc_saddle *refined;
    int n;
    uint8_t *active;
    board_cell *cells; /* insertion-ordered map BoardIdx -> Option<[usize;4]> */
    int n_cells, cap_cells;
    float spacing_ratio;
    unsigned score;
    nn_t *scratch;
} board_t;

static int board_find(const board_t *b, int x, int y)
{
    for (int i = 0; i < b->n_cells; ++i)
        if (b->cells[i].x == x && b->cells[i].y == y) return i;
    return -1;
}

static void board_insert(board_t *b, int x, int y, int has, const int *q)
{
    int i = board_find(b, x, y);
    if (i < 0) {
        if (b->n_cells == b->cap_cells) {
            b->cap_cells = b->cap_cells ? b->cap_cells * 2 : 64;
            b->cells = (board_cell *)realloc(b->cells, sizeof(board_cell) * (size_t)b->cap_cells);
        }
        i = b->n_cells++;
        b->cells[i].x = x;
        b->cells[i].y = y;
    }
    b->cells[i].has = has;
    if (has) memcpy(b->cells[i].q, q, sizeof(int) * 4);
}

/* find_closest_potential_saddle_idxs -- src/board.rs:177-233 */
static void board_closest(const board_t *b, const orc_saddle *s0, const orc_saddle *s1, int out0[3],
                          int *n0, int out1[3], int *n1)
{
    float ratio0 = 1.0f + b->spacing_ratio;
    float dx = s0->x - s1->x, dy = s0->y - s1->y;
    float radius_sq = 0.5f * (dx * dx + dy * dy);
    const float angle_thres = 5.0f;
    float v10x = s1->x - s0->x, v10y = s1->y - s0->y;
    float nv0x = s0->x + v10x * ratio0, nv0y = s0->y + v10y * ratio0;
    float nv1x = s1->x + v10x * ratio0, nv1y = s1->y + v10y * ratio0;
    nn_t nn[3];
    int m = knn(b->refined, b->n, nv0x, nv0y, 3, nn, b->scratch);
    *n0 = 0;
    for (int i = 0; i < m; ++i)
        if (nn[i].d <= radius_sq && b->active[nn[i].idx]) {
            float td = orc_theta_distance_degree(s0->theta, b->refined[nn[i].idx].theta);
            if (td < angle_thres) {
                out0[(*n0)++] = nn[i].idx;
                if (*n0 == 3) break;
            }
        }
    m = knn(b->refined, b->n, nv1x, nv1y, 3, nn, b->scratch);
    *n1 = 0;
    for (int i = 0; i < m; ++i)
        if (nn[i].d <= radius_sq && b->active[nn[i].idx]) {
            float td = orc_theta_distance_degree(s1->theta, b->refined[nn[i].idx].theta);
            if (td < angle_thres) {
                out1[(*n1)++] = nn[i].idx;
                if (*n1 == 3) break;
            }
        }
}

/* try_expand_one -- src/board.rs:153-176 */
static int board_expand_one(const board_t *b, const int q[4], int out[4])
{
    const orc_saddle *r = b->refined;
    int a0[3], a1[3], a2[3], a3[3], n0, n1, n2, n3;
    board_closest(b, &r[q[0]], &r[q[1]], a0, &n0, a1, &n1);
    board_closest(b, &r[q[3]], &r[q[2]], a3, &n3, a2, &n2);
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1)
            for (int i2 = 0; i2 < n2; ++i2)
                for (int i3 = 0; i3 < n3; ++i3)
                    if (orc_is_valid_quad(&r[a0[i0]], &r[a1[i1]], &r[a2[i2]], &r[a3[i3]])) {
                        out[0] = a0[i0]; out[1] = a1[i1]; out[2] = a2[i2]; out[3] = a3[i3];
                        return 1;
                    }
    return 0;
}

/* try_expand -- src/board.rs:114-152 */
static void board_expand(board_t *b, int bx, int by)
{
    int ci = board_find(b, bx, by);
    if (ci < 0 || !b->cells[ci].has) return;
    int quad[4];
    memcpy(quad, b->cells[ci].q, sizeof(quad));
    for (int i = 0; i < 4; ++i) {
        int qs[4];
        for (int j = 0; j < 4; ++j) qs[j] = quad[(j + i) & 3]; /* rotate_left(i) */
        int nx = bx, ny = by;
        if (i == 0) nx = bx + 1;
        else if (i == 1) ny = by - 1;
        else if (i == 2) nx = bx - 1;
        else ny = by + 1;
        int ni = board_find(b, nx, ny);
        if (ni >= 0 && b->cells[ni].has) continue;
        int nq[4];
        if (board_expand_one(b, qs, nq)) {
            int v[4];
            for (int j = 0; j < 4; ++j) v[(j + i) & 3] = nq[j]; /* rotate_right(i) */
            for (int j = 0; j < 4; ++j) b->active[v[j]] = 0;
            b->score += 1;
            board_insert(b, nx, ny, 1, v);
            board_expand(b, nx, ny);
        } else {
            board_insert(b, nx, ny, 0, NULL);
        }
    }
}

/* Board::new -- src/board.rs:27-48 */
static void board_new(board_t *b, const orc_saddle *refined, int n, const int quad[4],
                      float spacing_ratio, nn_t *scratch)
{
    memset(b, 0, sizeof(*b));
    b->refined = refined;
    b->n = n;
    b->active = (uint8_t *)malloc((size_t)n);
    memset(b->active, 1, (size_t)n);
    for (int i = 1; i < 4; ++i) b->active[quad[i]] = 0;
    b->spacing_ratio = spacing_ratio;
    b->score = 1;
    b->scratch = scratch;
    board_insert(b, 0, 0, 1, quad);
    board_expand(b, 0, 0);
}

static void board_free(board_t *b)
{
    free(b->active);
    free(b->cells);
    b->active = NULL;
    b->cells = NULL;
}

/* try_fix_missing -- src/board.rs:52-112 */
static void board_fix_missing(board_t *b)
{
    int n0 = b->n_cells;
    int (*fix)[4] = (int (*)[4])malloc(sizeof(int[4]) * (size_t)(n0 > 0 ? n0 : 1));
    int nfix = 0;
    for (int i = 0; i < n0; ++i) {
        const board_cell *c = &b->cells[i];
        if (c->has) continue;
        int i0 = board_find(b, c->x + 1, c->y), i1 = board_find(b, c->x - 1, c->y);
        int i2 = board_find(b, c->x, c->y + 1), i3 = board_find(b, c->x, c->y - 1);
        if (i0 >= 0 && i1 >= 0) {
            if (b->cells[i0].has && b->cells[i1].has) {
                fix[nfix][0] = c->x + 1; fix[nfix][1] = c->y; fix[nfix][2] = c->x - 1; fix[nfix][3] = c->y;
                nfix++;
            }
        } else if (i2 >= 0 && i3 >= 0 && b->cells[i2].has && b->cells[i3].has) {
            fix[nfix][0] = c->x; fix[nfix][1] = c->y + 1; fix[nfix][2] = c->x; fix[nfix][3] = c->y - 1;
            nfix++;
        }
    }
    for (int f = 0; f < nfix; ++f) {
        const board_cell *c0 = &b->cells[board_find(b, fix[f][0], fix[f][1])];
        const board_cell *c1 = &b->cells[board_find(b, fix[f][2], fix[f][3])];
        int q0[4], q1[4], sidx[4];
        memcpy(q0, c0->q, sizeof(q0));
        memcpy(q1, c1->q, sizeof(q1));
        for (int i = 0; i < 4; ++i) {
            float x = (b->refined[q0[i]].x + b->refined[q1[i]].x) / 2.0f;
            float y = (b->refined[q0[i]].y + b->refined[q1[i]].y) / 2.0f;
            nn_t nn;
            knn(b->refined, b->n, x, y, 1, &nn, b->scratch);
            sidx[i] = nn.idx;
        }
        if (orc_is_valid_quad(&b->refined[sidx[0]], &b->refined[sidx[1]], &b->refined[sidx[2]],
                              &b->refined[sidx[3]])) {
            /* integer division truncating toward zero, as Rust's i32 `/` */
            int bx = (fix[f][0] + fix[f][2]) / 2, by = (fix[f][1] + fix[f][3]) / 2;
            board_insert(b, bx, by, 1, sidx);
        }
    }
    free(fix);
}

/* init_quads -- src/detector.rs:543-586.  Appends quads to *quads (4 ints each). */
static void init_quads(const orc_saddle *refined, int n, int s0_idx, int **quads, int *nq, int *capq,
                       nn_t *scratch)
{
    nn_t nearest[50];
    const orc_saddle *s0 = &refined[s0_idx];
    int m = knn(refined, n, s0->x, s0->y, 50, nearest, scratch);
    int same[50], diff[50], ns = 0, nd = 0;
    for (int i = 1; i < m; ++i) {
        int s_idx = nearest[i].idx;
        float td = orc_theta_distance_degree(s0->theta, refined[s_idx].theta);
        if (td < 5.0f) same[ns++] = s_idx;
        else if (td > 80.0f) diff[nd++] = s_idx;
    }
    for (int si = 0; si < ns; ++si) {
        int s1_idx = same[si];
        const orc_saddle *s1 = &refined[s1_idx];
        for (int a = 0; a < nd; ++a)
            for (int b = a + 1; b < nd; ++b) { /* itertools combinations(2) order */
                const orc_saddle *d0 = &refined[diff[a]], *d1 = &refined[diff[b]];
                if (!orc_is_valid_quad(s0, d0, s1, d1)) continue;
                float c0 = orc_cross(d0->x - s0->x, d0->y - s0->y, s1->x - s0->x, s1->y - s0->y);
                if (*nq == *capq) {
                    *capq = *capq ? *capq * 2 : 64;
                    *quads = (int *)realloc(*quads, sizeof(int) * 4 * (size_t)*capq);
                }
                int *q = *quads + 4 * (*nq);
                q[0] = s0_idx;
                q[2] = s1_idx;
                if (c0 > 0.0f) { q[1] = diff[a]; q[3] = diff[b]; }
                else { q[1] = diff[b]; q[3] = diff[a]; }
                (*nq)++;
            }
    }
}

/* try_find_best_board -- src/detector.rs:588-639.
 * Writes up to cap quads (4 indices each) and returns their number, or -1 for None.
 * The reference picks the most populated round(theta) bin through a HashMap whose
 * iteration order is randomised per process, so ties between equally populated bins are
 * not deterministic there; here the tie goes to the smallest angle key.  Likewise
 * Board::all_tag_indexes iterates a HashMap: here cells come out in insertion order. */
int orc_try_find_best_board(const orc_saddle *refined, int n, int *out_quads, int cap)
{
    if (n <= 0) return -1;
    nn_t *scratch = (nn_t *)malloc(sizeof(nn_t) * (size_t)n);
    /* theta histogram, keys in [-90, 90] */
    int *bin_count = (int *)calloc(512, sizeof(int));
    for (int i = 0; i < n; ++i) {
        int angle = (int)roundf(refined[i].theta);
        bin_count[clampi(angle + 256, 0, 511)]++;
    }
    int best_bin = -1, best_len = 0;
    for (int b = 0; b < 512; ++b)
        if (bin_count[b] > best_len) { best_len = bin_count[b]; best_bin = b; }
    free(bin_count);
    int *s0_idxs = (int *)malloc(sizeof(int) * (size_t)best_len);
    int ns0 = 0;
    for (int i = 0; i < n; ++i)
        if (clampi((int)roundf(refined[i].theta) + 256, 0, 511) == best_bin) s0_idxs[ns0++] = i;

    unsigned best_score = 0;
    board_t best;
    int have_best = 0;
    int count = 0;
    int *quads = NULL, nq = 0, capq = 0;
    while (ns0 > 0 && count < 30) {
        int s0_idx = s0_idxs[--ns0];
        nq = 0;
        init_quads(refined, n, s0_idx, &quads, &nq, &capq, scratch);
        for (int qi = 0; qi < nq; ++qi) {
            board_t b;
            board_new(&b, refined, n, quads + 4 * qi, 0.3f, scratch);
            if (b.score > best_score) {
                if (have_best) board_free(&best);
                best = b;
                best_score = b.score;
                have_best = 1;
            } else {
                board_free(&b);
            }
        }
        if (best_score >= 36) break;
        count++;
    }
    free(quads);
    free(s0_idxs);
    int ret = -1;
    if (have_best) {
        board_fix_missing(&best);
        ret = 0;
        for (int i = 0; i < best.n_cells; ++i)
            if (best.cells[i].has) {
                if (ret < cap) memcpy(out_quads + 4 * ret, best.cells[i].q, sizeof(int) * 4);
                ret++;
            }
        board_free(&best);
    }
    free(scratch);
    return ret;
}

/* ---- decode ----------------------------------------------------------------------- */

/* Rust `f32 as u32`: saturating, NaN -> 0 */
static inline uint32_t f32_as_u32(float v)
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

/* tag_affine -- src/image_util.rs:39-70.  The reference solves the 8x6 system with
 * faer's f32 QR least squares; the system decouples into two 4x3 problems sharing the
 * design [sx, sy, 1].  Restated as normal equations in binary64, rounded once to
 * binary32 (faer's own rounding sequence is unpinned).  h = [h0..h5]. */
void orc_tag_affine(const float *corners_xy /* 4 pairs */, int side_bits, float margin, float *h)
{
    float S = (float)side_bits - 1.0f + margin;
    float src[4][2] = {{-margin, -margin}, {-margin, S}, {S, S}, {S, -margin}};
    double N[3][3] = {{0}}, bx[3] = {0}, by[3] = {0};
    for (int p = 0; p < 4; ++p) {
        double row[3] = {src[p][0], src[p][1], 1.0};
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) N[i][j] += row[i] * row[j];
            bx[i] += row[i] * corners_xy[2 * p];
            by[i] += row[i] * corners_xy[2 * p + 1];
        }
    }
    /* solve N a = bx, N b = by by Gauss-Jordan */
    double M[3][5];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) M[i][j] = N[i][j];
        M[i][3] = bx[i];
        M[i][4] = by[i];
    }
    for (int col = 0; col < 3; ++col) {
        int piv = col;
        for (int r = col + 1; r < 3; ++r)
            if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
        if (piv != col)
            for (int j = 0; j < 5; ++j) {
                double t = M[col][j]; M[col][j] = M[piv][j]; M[piv][j] = t;
            }
        double d = M[col][col];
        for (int j = 0; j < 5; ++j) M[col][j] /= d;
        for (int r = 0; r < 3; ++r)
            if (r != col) {
                double f = M[r][col];
                for (int j = 0; j < 5; ++j) M[r][j] -= f * M[col][j];
            }
    }
    h[0] = (float)M[0][3]; h[1] = (float)M[1][3]; h[2] = (float)M[2][3];
    h[3] = (float)M[0][4]; h[4] = (float)M[1][4]; h[5] = (float)M[2][4];
}

/* decode_positions -- src/detector.rs:42-72.  Returns 0 for None, else writes
 * edge*edge points. */
int orc_decode_positions(uint32_t img_w, uint32_t img_h, const float *quad_xy, int border_bits,
                         int edge_bits, float margin, float *out_xy)
{
    for (int i = 0; i < 4; ++i) {
        uint32_t x = f32_as_u32(roundf(quad_xy[2 * i]));
        uint32_t y = f32_as_u32(roundf(quad_xy[2 * i + 1]));
        if (x >= img_w || y >= img_h) return 0;
    }
    int side_bits = border_bits * 2 + edge_bits;
    float h[6];
    orc_tag_affine(quad_xy, side_bits, margin, h);
    int k = 0;
    for (int x = border_bits; x < border_bits + edge_bits; ++x)
        for (int y = border_bits; y < border_bits + edge_bits; ++y) {
            float fx = (float)x, fy = (float)y;
            /* 3x3 * 3x1 product, row dot products accumulated left to right */
            out_xy[2 * k] = h[0] * fx + h[1] * fy + h[2] * 1.0f;
            out_xy[2 * k + 1] = h[3] * fx + h[4] * fy + h[5] * 1.0f;
            k++;
        }
    return 1;
}

/* bit_code -- src/detector.rs:74-122.  Returns 1 and writes *bits, or 0 for None. */
int orc_bit_code(const uint8_t *img, uint32_t w, uint32_t h, const float *pts_xy, int n_pts,
                 int valid_brightness_threshold, uint32_t max_invalid_bit, uint64_t *bits_out)
{
    uint8_t bv[64];
    if (n_pts > 64) return 0;
    for (int i = 0; i < n_pts; ++i) {
        uint32_t x = f32_as_u32(roundf(pts_xy[2 * i])), y = f32_as_u32(roundf(pts_xy[2 * i + 1]));
        if (x >= w || y >= h) return 0; /* filter_map drops it -> length mismatch -> None */
        bv[i] = img[(size_t)y * w + x];
    }
    int min_b = 255, max_b = 0;
    for (int i = 0; i < n_pts; ++i) {
        if (bv[i] < min_b) min_b = bv[i];
        if (bv[i] > max_b) max_b = bv[i];
    }
    if (max_b - min_b < 50) return 0;
    int mid_b = (int)(uint8_t)f32_as_u32(roundf(((float)min_b + (float)max_b) / 2.0f));
    uint64_t bits = 0;
    uint32_t invalid = 0;
    for (int i = 0; i < n_pts; ++i) { /* .rev().enumerate(): i-th from the end -> bit i */
        int b = bv[n_pts - 1 - i];
        if (abs(mid_b - b) < valid_brightness_threshold) invalid++;
        if (b > mid_b) bits |= (uint64_t)1 << i;
    }
    if (invalid > max_invalid_bit) return 0;
    *bits_out = bits;
    return 1;
}

/* rotate_bits -- src/detector.rs:124-140 */
uint64_t orc_rotate_bits(uint64_t bits, int edge_bits)
{
    uint64_t b = 0;
    int count = 0;
    for (int r = edge_bits - 1; r >= 0; --r)
        for (int c = 0; c < edge_bits; ++c) {
            int idx = r + c * edge_bits;
            b |= ((bits >> idx) & 1u) << count;
            count++;
        }
    return b;
}

/* best_tag -- src/detector.rs:142-169.  Returns 1 and writes (idx, rotation), or 0. */
int orc_best_tag(uint64_t bits, int thres, const uint64_t *family, int n_codes, int edge_bits,
                 int *best_idx_out, int *rot_out)
{
    for (int rotated = 0; rotated < 4; ++rotated) {
        int best_idx = 0;
        unsigned best_score = (unsigned)__builtin_popcountll(family[0] ^ bits);
        for (int i = 1; i < n_codes; ++i) {
            unsigned s = (unsigned)__builtin_popcountll(family[i] ^ bits);
            if (s < best_score) { best_score = s; best_idx = i; }
        }
        if (best_score < (unsigned)thres) {
            *best_idx_out = best_idx;
            *rot_out = rotated;
            return 1;
        } else if (rotated == 3) {
            break;
        }
        bits = orc_rotate_bits(bits, edge_bits);
    }
    return 0;
}

/* try_decode_quad -- src/detector.rs:448-476 */
static int try_decode_quad(const uint8_t *grey, uint32_t w, uint32_t h, const float *quad_xy,
                           int border, int edge, int hamming, const uint64_t *codes, int n_codes,
                           int *tag_id, float *out_xy)
{
    float pts[2 * 64];
    if (!orc_decode_positions(w, h, quad_xy, border, edge, 0.5f, pts)) return 0;
    uint64_t bits;
    if (!orc_bit_code(grey, w, h, pts, edge * edge, 10, 3, &bits)) return 0;
    int idx, rot;
    if (!orc_best_tag(bits, hamming, codes, n_codes, edge, &idx, &rot)) return 0;
    float tmp[8];
    for (int i = 0; i < 4; ++i) { /* rotate_left(rot) */
        tmp[2 * i] = quad_xy[2 * ((i + rot) & 3)];
        tmp[2 * i + 1] = quad_xy[2 * ((i + rot) & 3) + 1];
    }
    for (int i = 0; i < 4; ++i) { /* reverse */
        out_xy[2 * i] = tmp[2 * (3 - i)];
        out_xy[2 * i + 1] = tmp[2 * (3 - i) + 1];
    }
    *tag_id = idx;
    return 1;
}

/* The host tail of TagDetector::detect -- src/detector.rs:510-539 -- from a saddle list
 * and the u8 luma plane.  `refined` is consumed (compacted in place).  Tags are written in
 * insertion order; a repeated id overwrites the earlier entry (HashMap::insert).
 * Returns the number of distinct tags (may exceed cap; only cap written). */
int orc_detect_tail(const uint8_t *grey, int w, int h, orc_saddle *refined, int n_refined,
                    int border, int edge, int hamming, const uint64_t *codes, int n_codes,
                    int max_num_of_boards, orc_tag *out, int cap)
{
    int n_tags = 0;
    int cap_q = n_refined > 16 ? n_refined : 16;
    int *quads = (int *)malloc(sizeof(int) * 4 * (size_t)cap_q);
    uint8_t *remove = (uint8_t *)malloc((size_t)(n_refined > 0 ? n_refined : 1));
    for (int board = 0; board < max_num_of_boards; ++board) {
        int nq = orc_try_find_best_board(refined, n_refined, quads, cap_q);
        if (nq < 0) continue;
        if (nq > cap_q) nq = cap_q;
        memset(remove, 0, (size_t)(n_refined > 0 ? n_refined : 1));
        for (int qi = 0; qi < nq; ++qi) {
            const int *q = quads + 4 * qi;
            float qxy[8];
            for (int i = 0; i < 4; ++i) {
                qxy[2 * i] = refined[q[i]].x;
                qxy[2 * i + 1] = refined[q[i]].y;
            }
            int id;
            float cxy[8];
            if (try_decode_quad(grey, (uint32_t)w, (uint32_t)h, qxy, border, edge, hamming, codes,
                                n_codes, &id, cxy)) {
                int slot = -1;
                for (int t = 0; t < n_tags && t < cap; ++t)
                    if (out[t].id == (uint32_t)id) { slot = t; break; }
                if (slot < 0) {
                    slot = n_tags < cap ? n_tags : -1;
                    n_tags++;
                }
                if (slot >= 0) {
                    out[slot].id = (uint32_t)id;
                    memcpy(out[slot].xy, cxy, sizeof(cxy));
                }
                for (int i = 0; i < 4; ++i) remove[q[i]] = 1;
            }
        }
        int m = 0;
        for (int i = 0; i < n_refined; ++i)
            if (!remove[i]) refined[m++] = refined[i];
        n_refined = m;
    }
    free(quads);
    free(remove);
    return n_tags;
}

/* TagDetector::detect -- src/detector.rs:505-540 (TagDetector::new's family table,
 * :364-406, is passed in by the caller as border / edge / hamming / codes). */
int orc_detect(const void *pixels, int w, int h, long stride_bytes, int fmt, const orc_params *prm,
               int border, int edge, int hamming, const uint64_t *codes, int n_codes, orc_tag *out,
               int cap)
{
    if (w < 2 || h < 2) return -2;
    uint8_t *grey = (uint8_t *)malloc((size_t)w * h);
    if (orc_luma_u8(pixels, w, h, stride_bytes, fmt, grey)) { free(grey); return -1; }
    int cap_s = 1 << 16;
    orc_saddle *refined = (orc_saddle *)malloc(sizeof(orc_saddle) * (size_t)cap_s);
    int ns = orc_refined_saddle_points(pixels, w, h, stride_bytes, fmt, prm, refined, cap_s, NULL);
    if (ns < 0) { free(grey); free(refined); return ns; }
    if (ns > cap_s) ns = cap_s;
    int nt = orc_detect_tail(grey, w, h, refined, ns, border, edge, hamming, codes, n_codes,
                             prm->max_num_of_boards, out, cap);
    free(grey);
    free(refined);
    return nt;
}

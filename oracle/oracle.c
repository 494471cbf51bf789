/*
 * oracle.c -- CPU restatement of the damaged-pixel detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it is
 * the checker the HIP path is compared against (tests/, __graft_entry__.smoke(),
 * and bench.py's cpu_baseline leg).  The product (image_processing_amd/) never
 * imports, links or executes this file.
 *
 * Reference citations are to /root/reference/image_processing.py ("ip.py").
 * The cv2 / numba arithmetic the reference delegates to third-party wheels
 * (opencv-python, numba: both unpinned, both absent from this image) is restated
 * here from the published algorithms; SURVEY.md App. A is the spec.  Parity
 * status per function is recorded in DESIGN.md ("pinned" = checked against the
 * reference's own known-answer tests or its own Python code run in the build
 * container; "parity unpinned" = no fixture of the reference covers it).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, optional -fopenmp).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <float.h>

#define EXPORT __attribute__((visibility("default")))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
/* cvRound: round-half-to-even (lrint in the default rounding mode). */
static inline int cv_round(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { return (int)floor(v); }

/* OpenCV borderInterpolate(p, len, BORDER_REFLECT_101) */
static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* ------------------------------------------------------------------------- */
/* _find_background, ip.py:203-252, numba typing per SURVEY App. A.1          */
/* frames: (n,h,w) float32, out bg: (h,w) float32                             */
/* ------------------------------------------------------------------------- */
EXPORT void orc_find_background_f32(const float *fr, int n, int h, int w,
                                    double coeff, float *bg)
{
    const size_t P = (size_t)h * (size_t)w;
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < P; ++p) {
        double s = 0.0;                                  /* ip.py:218-220 */
        for (int k = 0; k < n; ++k) s += (double)fr[(size_t)k * P + p];
        const float mean = (float)(s / (double)n);       /* ip.py:221 f32 store */
        s = 0.0;                                         /* ip.py:226-229 */
        for (int k = 0; k < n; ++k) {
            const float diff = fr[(size_t)k * P + p] - mean;   /* f32 - f32 */
            const float d2 = diff * diff;                      /* f32 * f32 */
            s += (double)d2;
        }
        const float sd = (float)sqrt(s / (double)n);     /* ip.py:230 */
        const double thr = (double)mean + coeff * (double)sd;   /* ip.py:235 */
        double si = 0.0;
        long c = 0;
        for (int k = 0; k < n; ++k) {                    /* ip.py:238-242 */
            const float v = fr[(size_t)k * P + p];
            if ((double)v <= thr) { si += (double)v; ++c; }
        }
        if (c > 0) {
            bg[p] = (float)(si / (double)c);             /* ip.py:244 */
        } else {
            double sa = 0.0;                             /* ip.py:247-250 */
            for (int k = 0; k < n; ++k) sa += (double)fr[(size_t)k * P + p];
            bg[p] = (float)(sa / (double)n);
        }
    }
}

/* ------------------------------------------------------------------------- */
/* cv2.blur(src64, (kw,kh)) == normalised boxFilter, anchor (kw/2,kh/2),     */
/* BORDER_REFLECT_101 (ip.py:643-644).  RowSum<double> then  */
/* ColumnSum<double> running sums, one multiply by 1/(kw*kh).                 */
/* ------------------------------------------------------------------------- */
EXPORT void orc_blur_f64(const double *src, int h, int w, int kh, int kw, double *dst)
{
    if (h == 1) kh = 1;
    if (w == 1) kw = 1;
    const int ax = kw / 2, ay = kh / 2;
    const double scale = 1.0 / ((double)kw * (double)kh);
    double *rows = (double *)malloc(sizeof(double) * (size_t)h * (size_t)w);
    /* horizontal running sum of every source row */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const double *S = src + (size_t)y * w;
        double *D = rows + (size_t)y * w;
        double s = 0.0;
        for (int i = 0; i < kw; ++i) s += S[reflect101(i - ax, w)];
        D[0] = s;
        for (int x = 1; x < w; ++x) {
            s += S[reflect101(x - ax + kw - 1, w)] - S[reflect101(x - 1 - ax, w)];
            D[x] = s;
        }
    }
    /* vertical running sum (ColumnSum): SUM holds kh-1 rows, add one, emit, drop one */
#pragma omp parallel for schedule(static)
    for (int x = 0; x < w; ++x) {
        double sum = 0.0;
        for (int i = 0; i < kh - 1; ++i) sum += rows[(size_t)reflect101(i - ay, h) * w + x];
        for (int y = 0; y < h; ++y) {
            const double s0 = sum + rows[(size_t)reflect101(y - ay + kh - 1, h) * w + x];
            dst[(size_t)y * w + x] = s0 * scale;
            sum = s0 - rows[(size_t)reflect101(y - ay, h) * w + x];
        }
    }
    free(rows);
}

/* Exact integer 30x30-style window sum for u8 frames (same window/border as above). */
EXPORT void orc_boxsum_u8(const uint8_t *src, int h, int w, int kh, int kw, int32_t *dst)
{
    if (h == 1) kh = 1;
    if (w == 1) kw = 1;
    const int ax = kw / 2, ay = kh / 2;
    int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)h * (size_t)w);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const uint8_t *S = src + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            int32_t s = 0;
            for (int i = 0; i < kw; ++i) s += S[reflect101(x - ax + i, w)];
            rows[(size_t)y * w + x] = s;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int32_t s = 0;
            for (int i = 0; i < kh; ++i) s += rows[(size_t)reflect101(y - ay + i, h) * w + x];
            dst[(size_t)y * w + x] = s;
        }
    free(rows);
}

/* ------------------------------------------------------------------------- */
/* connectedComponentsWithStats(mask, connectivity=8) (ip.py:672,746,757).    */
/* Only the partition and the areas are consumed; labels    */
/* here are numbered 1.. in raster order of each component's first pixel.     */
/* Returns num_labels (background included).                                  */
/* ------------------------------------------------------------------------- */
static int32_t uf_find(int32_t *p, int32_t x)
{
    while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; }
    return x;
}
static void uf_union(int32_t *p, int32_t a, int32_t b)
{
    a = uf_find(p, a); b = uf_find(p, b);
    if (a < b) p[b] = a; else if (b < a) p[a] = b;
}

EXPORT int orc_ccl8(const uint8_t *m, int h, int w, int32_t *labels)
{
    const size_t P = (size_t)h * w;
    int32_t *par = (int32_t *)malloc(sizeof(int32_t) * (P + 1));
    for (size_t i = 0; i < P; ++i) par[i] = (int32_t)i;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            if (!m[i]) continue;
            if (x > 0 && m[i - 1]) uf_union(par, (int32_t)i, (int32_t)(i - 1));
            if (y > 0) {
                if (m[i - w]) uf_union(par, (int32_t)i, (int32_t)(i - w));
                if (x > 0 && m[i - w - 1]) uf_union(par, (int32_t)i, (int32_t)(i - w - 1));
                if (x + 1 < w && m[i - w + 1]) uf_union(par, (int32_t)i, (int32_t)(i - w + 1));
            }
        }
    int next = 1;
    for (size_t i = 0; i < P; ++i) {
        if (!m[i]) { labels[i] = 0; continue; }
        const int32_t r = uf_find(par, (int32_t)i);
        if ((size_t)r == i) labels[i] = next++;   /* root = raster-first pixel */
        else labels[i] = labels[r];
    }
    free(par);
    return next;
}

/* ------------------------------------------------------------------------- */
/* findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE) outer border of the          */
/* component whose raster-first pixel is (x0,y0) (ip.py:691; Suzuki-Abe border */
/* following as in OpenCV's contour fetcher).  Directions:   */
/* 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE (image y grows downwards).             */
/* pts (optional, capacity cap points as x,y int32 pairs) receives the chain. */
/* ------------------------------------------------------------------------- */
EXPORT int orc_outer_contour(const uint8_t *img, int h, int w, int x0, int y0,
                             int32_t *pts, int cap, int *n_axis, int *n_diag)
{
    static const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    static const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
#define PIX(x, y) (((x) >= 0 && (x) < w && (y) >= 0 && (y) < h) ? img[(size_t)(y) * w + (x)] : 0)
    int s = 4, s_end = 4, i1x = x0, i1y = y0;
    int na = 0, nd = 0, np = 0;
    do {
        s = (s - 1) & 7;
        i1x = x0 + dx[s]; i1y = y0 + dy[s];
    } while (PIX(i1x, i1y) == 0 && s != s_end);
    if (s == s_end) {                       /* single-pixel domain */
        if (pts && cap > 0) { pts[0] = x0; pts[1] = y0; }
        *n_axis = 0; *n_diag = 0;
        return 1;
    }
    int i3x = x0, i3y = y0, i4x = x0, i4y = y0;
    for (;;) {
        while (s < 15) {
            ++s;
            i4x = i3x + dx[s & 7]; i4y = i3y + dy[s & 7];
            if (PIX(i4x, i4y) != 0) break;
        }
        s &= 7;
        if (pts && np < cap) { pts[2 * np] = i3x; pts[2 * np + 1] = i3y; }
        ++np;
        if (s & 1) ++nd; else ++na;
        if (i4x == x0 && i4y == y0 && i3x == i1x && i3y == i1y) break;
        i3x = i4x; i3y = i4y;
        s = (s + 4) & 7;
    }
#undef PIX
    *n_axis = na; *n_diag = nd;
    return np;
}

/* ========================================================================= */
/* Farneback dense optical flow (cv2.calcOpticalFlowFarneback, ip.py:858-867)  */
/* restated from the published OpenCV algorithm (modules/video optflowgf):    */
/* PARITY UNPINNED -- only t.py:706-727 (identical frames -> 0, different ->   */
/* >0) pins it; opencv-python is not installable here.                        */
/* ========================================================================= */

/* getGaussianKernel(n, sigma, CV_32F): double evaluation, double normalisation,
 * float storage; the fixed small table for sigma<=0. */
EXPORT void orc_gaussian_kernel_f32(int n, double sigma, float *k)
{
    static const double small3[3] = {0.25, 0.5, 0.25};
    static const double small5[5] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double small7[7] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    if (sigma <= 0 && (n == 1 || n == 3 || n == 5 || n == 7)) {
        if (n == 1) { k[0] = 1.f; return; }
        const double *t = n == 3 ? small3 : n == 5 ? small5 : small7;
        for (int i = 0; i < n; ++i) k[i] = (float)t[i];
        return;
    }
    const double sg = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2 = -0.5 / (sg * sg);
    double tmp[64];
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        tmp[i] = exp(scale2 * x * x);
        sum += tmp[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(tmp[i] * sum);
}

/* GaussianBlur(f32, (n,n), sigma), BORDER_REFLECT_101.  Separable, float
 * arithmetic, row pass then column pass.  Row pass: symmetric pairing for
 * n<=5 (small symmetric row filter), left-to-right tap sum otherwise; column
 * pass: symmetric pairing from the centre outwards. */
EXPORT void orc_gaussian_blur_f32(const float *src, int h, int w, int n, double sigma, float *dst)
{
    int nx = n, ny = n;
    if (w == 1) nx = 1;
    if (h == 1) ny = 1;
    float kx[64], ky[64];
    orc_gaussian_kernel_f32(nx, sigma, kx);
    orc_gaussian_kernel_f32(ny, sigma, ky);
    const int rx = nx / 2, ry = ny / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)h * w);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *S = src + (size_t)y * w;
        float *D = tmp + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            float s;
            if (nx <= 5) {
                s = S[x] * kx[rx];
                for (int j = 1; j <= rx; ++j)
                    s += (S[reflect101(x - j, w)] + S[reflect101(x + j, w)]) * kx[rx + j];
            } else {
                s = S[reflect101(x - rx, w)] * kx[0];
                for (int i = 1; i < nx; ++i) s += S[reflect101(x - rx + i, w)] * kx[i];
            }
            D[x] = s;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        float *D = dst + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            float s = ky[ry] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= ry; ++j)
                s += ky[ry + j] * (tmp[(size_t)reflect101(y + j, h) * w + x] +
                                   tmp[(size_t)reflect101(y - j, h) * w + x]);
            D[x] = s;
        }
    }
    free(tmp);
}

/* cv::resize(src, dst, Size(dw,dh), INTER_LINEAR) for float, cn interleaved
 * channels.  Exact 2x2 decimation is taken by the area-average path (OpenCV
 * maps INTER_LINEAR to INTER_AREA when both integer scales are 2); same size is
 * a copy; otherwise horizontal then vertical 2-tap interpolation in float. */
EXPORT void orc_resize_linear_f32(const float *src, int sh, int sw, int cn,
                                  float *dst, int dh, int dw)
{
    if (sh == dh && sw == dw) {
        memcpy(dst, src, sizeof(float) * (size_t)sh * sw * cn);
        return;
    }
    if (sw == dw * 2 && sh == dh * 2) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < dh; ++y) {
            const float *S0 = src + (size_t)(2 * y) * sw * cn;
            const float *S1 = S0 + (size_t)sw * cn;
            float *D = dst + (size_t)y * dw * cn;
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < cn; ++c)
                    D[x * cn + c] = (S0[(2 * x) * cn + c] + S0[(2 * x + 1) * cn + c] +
                                     S1[(2 * x) * cn + c] + S1[(2 * x + 1) * cn + c]) * 0.25f;
        }
        return;
    }
    const double scale_x = 1.0 / ((double)dw / sw), scale_y = 1.0 / ((double)dh / sh);
    int *xo = (int *)malloc(sizeof(int) * dw);
    float *xa = (float *)malloc(sizeof(float) * dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xo[dx] = sx; xa[dx] = fx;
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        const int y0 = imin(imax(sy, 0), sh - 1), y1 = imin(imax(sy + 1, 0), sh - 1);
        const float b0 = 1.f - fy, b1 = fy;
        const float *S0 = src + (size_t)y0 * sw * cn, *S1 = src + (size_t)y1 * sw * cn;
        float *D = dst + (size_t)dy * dw * cn;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xo[dx];
            const float a1 = xa[dx], a0 = 1.f - a1;
            for (int c = 0; c < cn; ++c) {
                float h0, h1;
                if (sx >= sw - 1) {          /* right edge: single tap times one */
                    h0 = S0[sx * cn + c];
                    h1 = S1[sx * cn + c];
                } else {
                    h0 = S0[sx * cn + c] * a0 + S0[(sx + 1) * cn + c] * a1;
                    h1 = S1[sx * cn + c] * a0 + S1[(sx + 1) * cn + c] * a1;
                }
                D[dx * cn + c] = h0 * b0 + h1 * b1;
            }
        }
    }
    free(xo); free(xa);
}

/* FarnebackPrepareGaussian: taps g, x*g, x*x*g (float) and the four inverse
 * Gram entries (double).  The 6x6 Gram matrix is inverted in closed form: it
 * decouples into {1,x^2,y^2}, {x}, {y}, {xy} blocks. */
static void fb_prepare(int n, double sigma, float *g, float *xg, float *xxg,
                       double *ig11, double *ig03, double *ig33, double *ig55)
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.0;
    for (int x = -n; x <= n; ++x) {
        g[x] = (float)exp(-x * x / (2 * sigma * sigma));
        s += g[x];
    }
    s = 1.0 / s;
    for (int x = -n; x <= n; ++x) {
        g[x] = (float)(g[x] * s);
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G00 = 0, G11 = 0, G33 = 0, G55 = 0;
    for (int y = -n; y <= n; ++y)
        for (int x = -n; x <= n; ++x) {
            G00 += g[y] * g[x];
            G11 += g[y] * g[x] * x * x;
            G33 += g[y] * g[x] * x * x * x * x;
            G55 += g[y] * g[x] * x * x * y * y;
        }
    /* G = [[G00,0,0,G11,G11,0],[0,G11,..],[..G11..],[G11,0,0,G33,G55,0],[G11,0,0,G55,G33,0],[..G55]] */
    /* inverse of the 3x3 block A = [[a,b,b],[b,c,d],[b,d,c]], a=G00,b=G11,c=G33,d=G55 */
    const double a = G00, b = G11, c = G33, d = G55;
    const double det = a * (c * c - d * d) - 2.0 * b * b * (c - d);
    *ig11 = 1.0 / G11;
    *ig03 = (b * d - b * c) / det;           /* cofactor(0,1)/det, symmetric */
    *ig33 = (a * c - b * b) / det;
    *ig55 = 1.0 / G55;
}

/* FarnebackPolyExp: I (h,w) f32 -> R (h,w,5) f32 */
EXPORT void orc_polyexp(const float *src, int height, int width, int n, double sigma, float *dst)
{
    float kbuf[3 * 64];
    float *g = kbuf + n, *xg = g + 2 * n + 1, *xxg = xg + 2 * n + 1;
    double ig11, ig03, ig33, ig55;
    fb_prepare(n, sigma, g, xg, xxg, &ig11, &ig03, &ig33, &ig55);
#pragma omp parallel
    {
        float *rowbuf = (float *)malloc(sizeof(float) * (size_t)(width + 2 * n) * 3);
        float *row = rowbuf + n * 3;
#pragma omp for schedule(static)
        for (int y = 0; y < height; ++y) {
            float g0 = g[0], g1, g2;
            const float *srow0 = src + (size_t)y * width, *srow1;
            float *drow = dst + (size_t)y * width * 5;
            for (int x = 0; x < width; ++x) {
                row[x * 3] = srow0[x] * g0;
                row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
            }
            for (int k = 1; k <= n; ++k) {
                g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
                srow0 = src + (size_t)imax(y - k, 0) * width;
                srow1 = src + (size_t)imin(y + k, height - 1) * width;
                for (int x = 0; x < width; ++x) {
                    const float p = srow0[x] + srow1[x];
                    const float t0 = row[x * 3] + g0 * p;
                    const float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                    const float t2 = row[x * 3 + 2] + g2 * p;
                    row[x * 3] = t0; row[x * 3 + 1] = t1; row[x * 3 + 2] = t2;
                }
            }
            for (int x = 0; x < n * 3; ++x) {
                row[-1 - x] = row[2 - x];
                row[width * 3 + x] = row[width * 3 + x - 3];
            }
            for (int x = 0; x < width; ++x) {
                g0 = g[0];
                double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0,
                       b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
                for (int k = 1; k <= n; ++k) {
                    const double tg = row[(x + k) * 3] + row[(x - k) * 3];
                    g0 = g[k];
                    b1 += tg * g0; b4 += tg * xxg[k];
                    b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                    b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                    b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                    b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
                }
                drow[x * 5 + 1] = (float)(b2 * ig11);
                drow[x * 5] = (float)(b3 * ig11);
                drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
                drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
                drow[x * 5 + 4] = (float)(b6 * ig55);
            }
        }
        free(rowbuf);
    }
}

/* FarnebackUpdateMatrices over all rows: R0,R1 (h,w,5), flow (h,w,2) -> M (h,w,5) */
EXPORT void orc_update_matrices(const float *R0a, const float *R1, const float *flowa,
                                int height, int width, float *Ma)
{
    static const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    const int BORDER = 5;
    const size_t step1 = (size_t)width * 5;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; ++y) {
        const float *flow = flowa + (size_t)y * width * 2;
        const float *R0 = R0a + (size_t)y * width * 5;
        float *M = Ma + (size_t)y * width * 5;
        for (int x = 0; x < width; ++x) {
            const float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            const int x1 = cv_floor(fx), y1 = cv_floor(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= y1;
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1 < (unsigned)(height - 1)) {
                const float *ptr = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
                const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy),
                            a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
                const float scale = (x < BORDER ? border[x] : 1.f) *
                                    (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                                    (y < BORDER ? border[y] : 1.f) *
                                    (y >= height - BORDER ? border[height - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

/* FarnebackUpdateFlow_Blur's box window + 2x2 solve: M (h,w,5) -> flow (h,w,2).
 * Vertical then horizontal running sums in double, replicate borders.  (The
 * reference implementation's striped in-loop matrix update only touches rows the
 * blur no longer reads and uses final flow rows, so it equals a full update
 * after the pass; the caller does that.) */
EXPORT void orc_blur_solve(const float *Ma, int height, int width, int block_size, float *flowa)
{
    const int m = block_size / 2;
    const double scale = 1.0 / ((double)block_size * block_size);
    double *vbuf = (double *)malloc(sizeof(double) * (size_t)(width + m * 2 + 2) * 5);
    double *vsum = vbuf + (m + 1) * 5;
    const float *srow0 = Ma;
    for (int x = 0; x < width * 5; ++x) vsum[x] = srow0[x] * (m + 2);
    for (int y = 1; y < m; ++y) {
        srow0 = Ma + (size_t)imin(y, height - 1) * width * 5;
        for (int x = 0; x < width * 5; ++x) vsum[x] += srow0[x];
    }
    for (int y = 0; y < height; ++y) {
        float *flow = flowa + (size_t)y * width * 2;
        srow0 = Ma + (size_t)imax(y - m - 1, 0) * width * 5;
        const float *srow1 = Ma + (size_t)imin(y + m, height - 1) * width * 5;
        for (int x = 0; x < width * 5; ++x) vsum[x] += srow1[x] - srow0[x];
        for (int x = 0; x < (m + 1) * 5; ++x) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[width * 5 + x] = vsum[width * 5 + x - 5];
        }
        double g11 = vsum[0] * (m + 2), g12 = vsum[1] * (m + 2), g22 = vsum[2] * (m + 2),
               h1 = vsum[3] * (m + 2), h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; ++x) {
            g11 += vsum[x * 5]; g12 += vsum[x * 5 + 1]; g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3]; h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < width; ++x) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            const double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale,
                         h1_ = h1 * scale, h2_ = h2 * scale;
            const double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
    }
    free(vbuf);
}

/* Number of pyramid levels actually used and the size / blur of level k. */
EXPORT int orc_fb_levels(int h, int w, double pyr_scale, int levels)
{
    int k; double scale = 1;
    for (k = 0; k < levels; ++k) {
        scale *= pyr_scale;
        if (w * scale < 32 || h * scale < 32) break;
    }
    return k;
}
EXPORT void orc_fb_level_geom(int h, int w, double pyr_scale, int k,
                              int *lh, int *lw, double *sigma, int *smooth_sz)
{
    double scale = 1;
    for (int i = 0; i < k; ++i) scale *= pyr_scale;
    *sigma = (1. / scale - 1) * 0.5;
    int sz = cv_round(*sigma * 5) | 1;
    *smooth_sz = imax(sz, 3);
    *lw = cv_round(w * scale);
    *lh = cv_round(h * scale);
}

/* calcOpticalFlowFarneback(prev, next, None, pyr_scale, levels, winsize, iters,
 * poly_n, poly_sigma, flags=0): prev/next (h,w) f32 -> flow (h,w,2) f32. */
EXPORT void orc_farneback(const float *prev, const float *next, int h, int w,
                          double pyr_scale, int levels, int winsize, int iters,
                          int poly_n, double poly_sigma, float *flow0)
{
    const float *img[2] = {prev, next};
    levels = orc_fb_levels(h, w, pyr_scale, levels);
    float *prev_flow = NULL; int ph = 0, pw = 0;
    float *fimg = (float *)malloc(sizeof(float) * (size_t)h * w);
    for (int k = levels; k >= 0; --k) {
        int lh, lw, smooth_sz; double sigma;
        orc_fb_level_geom(h, w, pyr_scale, k, &lh, &lw, &sigma, &smooth_sz);
        const size_t LP = (size_t)lh * lw;
        float *flow = (k > 0) ? (float *)malloc(sizeof(float) * LP * 2) : flow0;
        if (!prev_flow) {
            memset(flow, 0, sizeof(float) * LP * 2);
        } else {
            orc_resize_linear_f32(prev_flow, ph, pw, 2, flow, lh, lw);
            const float mul = (float)(1. / pyr_scale);
            for (size_t i = 0; i < LP * 2; ++i) flow[i] *= mul;
        }
        float *R[2], *I = (float *)malloc(sizeof(float) * LP);
        float *M = (float *)malloc(sizeof(float) * LP * 5);
        for (int i = 0; i < 2; ++i) {
            orc_gaussian_blur_f32(img[i], h, w, smooth_sz, sigma, fimg);
            orc_resize_linear_f32(fimg, h, w, 1, I, lh, lw);
            R[i] = (float *)malloc(sizeof(float) * LP * 5);
            orc_polyexp(I, lh, lw, poly_n, poly_sigma, R[i]);
        }
        orc_update_matrices(R[0], R[1], flow, lh, lw, M);
        for (int i = 0; i < iters; ++i) {
            orc_blur_solve(M, lh, lw, winsize, flow);
            if (i < iters - 1) orc_update_matrices(R[0], R[1], flow, lh, lw, M);
        }
        free(R[0]); free(R[1]); free(I); free(M);
        if (prev_flow) free(prev_flow);
        prev_flow = (k > 0) ? flow : NULL; ph = lh; pw = lw;
    }
    free(fimg);
}

/* cartToPolar magnitude in float: sqrt(x*x + y*y), ip.py:868 */
EXPORT void orc_flow_magnitude(const float *flow, size_t npix, float *mag)
{
    for (size_t i = 0; i < npix; ++i) {
        const float x = flow[2 * i], y = flow[2 * i + 1];
        mag[i] = sqrtf(x * x + y * y);
    }
}

/* cvtColor(BGR2GRAY) for u8, OpenCV 4.x 15-bit fixed point:
 * (B*3735 + G*19235 + R*9798 + 16384) >> 15.  PARITY UNPINNED. */
EXPORT void orc_bgr2gray_u8(const uint8_t *bgr, size_t npix, uint8_t *gray)
{
    for (size_t i = 0; i < npix; ++i) {
        const unsigned b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15);
    }
}

/* asan_driver.c -- `make -C oracle asan`: the C half of the CPU restatement (oracle.c) compiled together with this
 * driver under -fsanitize=address,undefined and run once over every entry point, on small inputs that include the
 * awkward shapes (1-pixel-wide images, sizes below the filter apertures, odd pyramid sizes).  TEST INFRASTRUCTURE ONLY,
 * like everything under oracle/.  The driver checks no values (tests/test_oracle_*.py do that): it exists so that an
 * out-of-bounds access, a misaligned or overflowing operation in the checker itself cannot go unnoticed (SURVEY
 * section 5).  Exit status 0 = the sanitizers saw nothing. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_find_background_f32(const float *fr, int n, int h, int w, double coeff, float *bg);
void orc_blur_f64(const double *src, int h, int w, int kh, int kw, double *dst);
void orc_boxsum_u8(const uint8_t *src, int h, int w, int kh, int kw, int32_t *dst);
int orc_ccl8(const uint8_t *m, int h, int w, int32_t *labels);
int orc_outer_contour(const uint8_t *img, int h, int w, int x0, int y0, int32_t *pts, int cap, int *n_axis, int *n_diag);
void orc_gaussian_kernel_f32(int n, double sigma, float *k);
void orc_gaussian_blur_f32(const float *src, int h, int w, int n, double sigma, float *dst);
void orc_resize_linear_f32(const float *src, int sh, int sw, int cn, float *dst, int dh, int dw);
void orc_polyexp(const float *src, int height, int width, int n, double sigma, float *dst);
void orc_update_matrices(const float *R0a, const float *R1, const float *flowa, int height, int width, float *Ma);
void orc_blur_solve(const float *Ma, int height, int width, int block_size, float *flowa);
int orc_fb_levels(int h, int w, double pyr_scale, int levels);
void orc_fb_level_geom(int h, int w, double pyr_scale, int k, int *lh, int *lw, double *sigma, int *smooth_sz);
void orc_farneback(const float *prev, const float *next, int h, int w, double pyr_scale, int levels, int winsize, int iters,
                   int poly_n, double poly_sigma, float *flow0);
void orc_flow_magnitude(const float *flow, size_t npix, float *mag);
void orc_bgr2gray_u8(const uint8_t *bgr, size_t npix, uint8_t *gray);

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

/* exact-size heap blocks, so that the sanitizer sees the first byte past every array */
static float *f32_bytes(size_t n) { float *p = malloc(sizeof(float) * (n ? n : 1)); for (size_t i = 0; i < n; ++i) p[i] = (float)(rnd() & 255); return p; }
static uint8_t *u8_bytes(size_t n, unsigned mod) { uint8_t *p = malloc(n ? n : 1); for (size_t i = 0; i < n; ++i) p[i] = (uint8_t)(rnd() % mod); return p; }

static void one_size(int h, int w)
{
    const size_t P = (size_t)h * w;
    /* temporal background over 1..6 neighbours */
    for (int n = 1; n <= 6; n += 2) {
        float *fr = f32_bytes(P * n), *bg = f32_bytes(P);
        orc_find_background_f32(fr, n, h, w, 1.0, bg);
        orc_find_background_f32(fr, n, h, w, 0.5, bg);
        free(fr); free(bg);
    }
    /* 30x30 mean / box sum with reflect-101 borders (also on images smaller than the window) */
    {
        uint8_t *u = u8_bytes(P, 256);
        double *d = malloc(sizeof(double) * P), *o = malloc(sizeof(double) * P);
        int32_t *s = malloc(sizeof(int32_t) * P);
        for (size_t i = 0; i < P; ++i) d[i] = u[i];
        orc_blur_f64(d, h, w, 30, 30, o);
        orc_blur_f64(d, h, w, 3, 3, o);
        orc_boxsum_u8(u, h, w, 30, 30, s);
        free(u); free(d); free(o); free(s);
    }
    /* labelling + outer contour of every component's raster-first pixel, sparse and dense masks */
    for (unsigned mod = 2; mod <= 5; mod += 3) {
        uint8_t *m = u8_bytes(P, mod);
        for (size_t i = 0; i < P; ++i) m[i] = m[i] == 0;
        int32_t *lab = malloc(sizeof(int32_t) * P);
        const int nl = orc_ccl8(m, h, w, lab);
        uint8_t *seen = calloc((size_t)nl + 1, 1), *one = malloc(P ? P : 1);
        int32_t *pts = malloc(sizeof(int32_t) * 2 * 64);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int l = lab[(size_t)y * w + x];
                if (l <= 0 || l > nl || seen[l]) continue;
                seen[l] = 1;
                for (size_t i = 0; i < P; ++i) one[i] = lab[i] == l;
                int na, nd;
                (void)orc_outer_contour(one, h, w, x, y, pts, 64, &na, &nd);
                (void)orc_outer_contour(one, h, w, x, y, NULL, 0, &na, &nd);
            }
        free(m); free(lab); free(seen); free(one); free(pts);
    }
    /* Farneback pieces and the whole thing */
    {
        float *a = f32_bytes(P), *b = f32_bytes(P), *t = f32_bytes(P);
        float k[32];
        orc_gaussian_kernel_f32(3, 0.0, k);
        orc_gaussian_kernel_f32(9, 1.5, k);
        orc_gaussian_kernel_f32(19, 3.5, k);
        orc_gaussian_blur_f32(a, h, w, 3, 0.0, t);
        orc_gaussian_blur_f32(a, h, w, 19, 3.5, t);
        const int dh = h / 2 > 0 ? h / 2 : 1, dw = w / 2 > 0 ? w / 2 : 1;
        float *half = f32_bytes((size_t)dh * dw), *up = f32_bytes(P * 2), *two = f32_bytes((size_t)dh * dw * 2);
        orc_resize_linear_f32(a, h, w, 1, half, dh, dw);
        orc_resize_linear_f32(two, dh, dw, 2, up, h, w);
        float *R0 = f32_bytes(P * 5), *R1 = f32_bytes(P * 5), *M = f32_bytes(P * 5), *flow = f32_bytes(P * 2), *mag = f32_bytes(P);
        orc_polyexp(a, h, w, 5, 2.0, R0);
        orc_polyexp(b, h, w, 5, 2.0, R1);
        for (size_t i = 0; i < 2 * P; ++i) flow[i] = (float)((int)(rnd() % 41) - 20) * 0.37f;    /* samples inside and outside */
        orc_update_matrices(R0, R1, flow, h, w, M);
        orc_blur_solve(M, h, w, 15, flow);
        int lh, lw, sz; double sg;
        const int lv = orc_fb_levels(h, w, 0.5, 3);
        for (int q = 0; q <= lv; ++q) orc_fb_level_geom(h, w, 0.5, q, &lh, &lw, &sg, &sz);
        orc_farneback(a, b, h, w, 0.5, 3, 15, 3, 5, 2.0, flow);
        orc_flow_magnitude(flow, P, mag);
        free(a); free(b); free(t); free(half); free(up); free(two); free(R0); free(R1); free(M); free(flow); free(mag);
    }
    /* luma */
    {
        uint8_t *bgr = u8_bytes(3 * P, 256), *g = u8_bytes(P, 256);
        orc_bgr2gray_u8(bgr, P, g);
        free(bgr); free(g);
    }
}

int main(void)
{
    static const int sizes[][2] = {{1, 1}, {1, 7}, {9, 1}, {2, 2}, {5, 31}, {33, 34}, {40, 56}, {67, 131}, {264, 360}};
    for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) one_size(sizes[i][0], sizes[i][1]);
    puts("oracle asan driver: ok");
    return 0;
}

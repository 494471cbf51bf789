/* A plain C (C99) caller of include/ipx.h: one chunk through ipx_detect_chunk on device 0 and through
 * ipx_group_detect_chunk with `ranks` ranks on device 0 (peer-copy transport), results compared.
 *   gcc -std=c99 -pedantic -Wall -Iinclude tests/c_abi/group_example.c -Limage_processing_amd/lib -lipx -lamdhip64 -lm -o group_example
 *   ./group_example <pow2_table.bin> <tdim> <ranks>
 * (the table of perimeter squares comes from the host binding, image_processing_amd/hostmath.py: pow2_table)
 * Device memory for the single-GPU call is taken with hipMalloc through the tiny prototypes below, so that the
 * example needs no HIP headers. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "ipx.h"

extern int hipMalloc(void **ptr, size_t bytes);
extern int hipFree(void *ptr);
extern int hipMemcpy(void *dst, const void *src, size_t bytes, int kind);   /* 1 = H2D, 2 = D2H */

#define CHECK(call) do { int s_ = (call); if (s_) { fprintf(stderr, "%s -> %d: %s\n", #call, s_, ipx_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int n = 24, H = 96, W = 128;
    const size_t P = (size_t)H * W;
    int tdim, ranks, i, x, y;
    size_t got;
    double *tab;
    unsigned char *frames;
    FILE *f;
    if (argc < 4) { fprintf(stderr, "usage: %s table.bin tdim ranks\n", argv[0]); return 2; }
    tdim = atoi(argv[2]); ranks = atoi(argv[3]);
    tab = (double *)malloc(sizeof(double) * tdim * tdim);
    f = fopen(argv[1], "rb");
    if (!f || !tab) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    got = fread(tab, sizeof(double), (size_t)tdim * tdim, f);
    fclose(f);
    if (got != (size_t)tdim * tdim) { fprintf(stderr, "short table\n"); return 2; }

    /* a synthetic chunk: smooth texture, a few hot pixels that come and go, one jump of the scene */
    frames = (unsigned char *)malloc(n * P);
    for (i = 0; i < n; ++i)
        for (y = 0; y < H; ++y)
            for (x = 0; x < W; ++x) {
                double xs = x - (i >= 13 ? 7.0 : 0.0);
                int v = (int)floor(90.5 + 50.0 * sin(xs / 9.0) * cos(y / 7.0) + 30.0 * sin((xs + 2.0 * y) / 13.0)) + ((x * 7 + y * 13 + i * 5) % 3);
                if ((x * 31 + y * 17) % 977 == (i % 3)) v = 250;
                frames[(size_t)i * P + (size_t)y * W + x] = (unsigned char)v;
            }

    {
        ipx_ctx *ctx;
        ipx_group *g;
        ipx_params p;
        void *d_frames, *d_tab, *d_flows, *d_keep, *d_counts, *d_cluster;
        float *flows1 = (float *)malloc(sizeof(float) * n), *flows2 = (float *)malloc(sizeof(float) * n);
        unsigned char *keep1 = (unsigned char *)malloc(n), *keep2 = (unsigned char *)malloc(n);
        int64_t *counts1 = (int64_t *)malloc(sizeof(int64_t) * 5 * n), *counts2 = (int64_t *)malloc(sizeof(int64_t) * 5 * n);
        int64_t *cluster1 = (int64_t *)malloc(sizeof(int64_t) * 3 * n), *cluster2 = (int64_t *)malloc(sizeof(int64_t) * 3 * n);
        int kept = 0;

        ipx_default_params(&p);
        CHECK(ipx_ctx_create(0, &ctx));
        CHECK(ipx_set_params(ctx, &p));
        if (hipMalloc(&d_frames, n * P) || hipMalloc(&d_tab, sizeof(double) * tdim * tdim) || hipMalloc(&d_flows, sizeof(float) * n) ||
            hipMalloc(&d_keep, n) || hipMalloc(&d_counts, sizeof(int64_t) * 5 * n) || hipMalloc(&d_cluster, sizeof(int64_t) * 3 * n)) {
            fprintf(stderr, "hipMalloc failed\n");
            return 1;
        }
        hipMemcpy(d_frames, frames, n * P, 1);
        hipMemcpy(d_tab, tab, sizeof(double) * tdim * tdim, 1);
        CHECK(ipx_detect_chunk(ctx, (const uint8_t *)d_frames, n, H, W, (const double *)d_tab, tdim, (float *)d_flows, (uint8_t *)d_keep,
                               (int64_t *)d_counts, (int64_t *)d_cluster, NULL));
        CHECK(ipx_synchronize(ctx));
        hipMemcpy(flows1, d_flows, sizeof(float) * n, 2);
        hipMemcpy(keep1, d_keep, n, 2);
        hipMemcpy(counts1, d_counts, sizeof(int64_t) * 5 * n, 2);
        hipMemcpy(cluster1, d_cluster, sizeof(int64_t) * 3 * n, 2);

        {
            int devs[16];
            for (i = 0; i < ranks && i < 16; ++i) devs[i] = 0;
            CHECK(ipx_group_create(devs, ranks, IPX_GROUP_PEER, &g));
        }
        CHECK(ipx_group_set_params(g, &p));
        CHECK(ipx_group_detect_chunk(g, frames, n, H, W, tab, tdim, flows2, keep2, counts2, cluster2));
        CHECK(ipx_group_destroy(g));

        if (memcmp(flows1, flows2, sizeof(float) * n) || memcmp(keep1, keep2, n) || memcmp(counts1, counts2, sizeof(int64_t) * 5 * n) ||
            memcmp(cluster1, cluster2, sizeof(int64_t) * 3 * n)) {
            fprintf(stderr, "group result differs from the single-context result\n");
            return 1;
        }
        for (i = 0; i < n; ++i) kept += keep1[i];
        printf("ok: %d frames, %d kept, %d ranks, damaged pixels of frame 1: %lld\n", n, kept, ranks, (long long)counts1[5]);
        hipFree(d_frames); hipFree(d_tab); hipFree(d_flows); hipFree(d_keep); hipFree(d_counts); hipFree(d_cluster);
        CHECK(ipx_ctx_destroy(ctx));
    }
    return 0;
}

// chunk.hip -- detect_damaged_pixels (ip.py:526-632) for one chunk resident in HBM: every stage
// is enqueued on the context's stream in the reference's order; nothing returns to the host
// until the caller synchronises.  The integer counters leave the GPU; the float glue of
// ip.py:605-613 (estimates, NaN rules, means) is done by the host binding from those integers.
#include "ipx_internal.h"

namespace {

// counts[i] = { sum(final), #(final & f < est), #(final & f > bt), #(f < est), #(f > bt) }
__global__ void merge_counts_kernel(const int64_t *fin3, const int64_t *dense2, int n, int64_t *out5)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out5[5 * i + 0] = fin3[3 * i + 0];
    out5[5 * i + 1] = fin3[3 * i + 1];
    out5[5 * i + 2] = fin3[3 * i + 2];
    out5[5 * i + 3] = dense2[2 * i + 0];
    out5[5 * i + 4] = dense2[2 * i + 1];
}

}  // namespace

// (internal: also the last step of a rank of ipx_group_detect_chunk, group.hip)
int ipx_merge_counts(ipx_ctx *ctx, const int64_t *fin3, const int64_t *dense2, int n, int64_t *out5)
{
    IPX_TRY(ipx_ctx_enter(ctx));
    IPX_LAUNCH(ctx, KID_MISC, 80.0 * n, merge_counts_kernel, dim3((n + 255) / 256), dim3(256), fin3, dense2, n, out5);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_detect_chunk(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W,
                                const double *pow2_table, int tdim,
                                float *flows, uint8_t *keep, int64_t *counts, int64_t *cluster,
                                uint64_t *final_bits)
{
    IPX_REQUIRE(ctx && frames && flows && keep && counts && cluster, "null pointer");
    IPX_REQUIRE(n >= 2 && n <= 65535 && H >= 2 && W >= 2, "bad sizes (2 <= n <= 65535 frames of at least 2x2)");
    IPX_TRY(ipx_ctx_enter(ctx));
    const size_t words = (size_t)H * ipx_wpr(W), P = (size_t)H * W;
    void *raw, *bright, *pers, *stat, *fin = final_bits, *maps, *cnt;
    IPX_TRY(ipx_scratch(ctx, WS_RAW, words * 8 * n, &raw));
    IPX_TRY(ipx_scratch(ctx, WS_BRIGHT, words * 8 * n, &bright));
    IPX_TRY(ipx_scratch(ctx, WS_PERS, words * 8, &pers));
    IPX_TRY(ipx_scratch(ctx, WS_STATIC, words * 8, &stat));
    if (!fin) IPX_TRY(ipx_scratch(ctx, WS_FINAL, words * 8 * n, &fin));
    IPX_TRY(ipx_scratch(ctx, WS_MAPS, P * 4 * 2, &maps));
    IPX_TRY(ipx_scratch(ctx, WS_COUNTS, sizeof(int64_t) * 5 * (size_t)n, &cnt));
    int64_t *dense2 = (int64_t *)cnt, *fin3 = dense2 + 2 * (size_t)n;
    int32_t *heat = (int32_t *)maps, *valid = heat + P;

    // (each stage's enqueue sits in a roctx range named after the reference function it replaces)
    ipx_range whole("ipx_detect_chunk");
    {
        ipx_range r("compute_optical_flow_metric");
        IPX_TRY(ipx_flow_metric(ctx, frames, n, H, W, nullptr, flows, keep));                     // ip.py:554, :559
    }
    {
        ipx_range r("compute_background+_raw_damaged_mask");
        IPX_TRY(ipx_detect_raw(ctx, frames, n, H, W, 0, n, keep, (uint64_t *)raw, (uint64_t *)bright, dense2));   // :560-565
    }
    {
        ipx_range r("remove_bright_regions");
        IPX_TRY(ipx_remove_bright_regions(ctx, (const uint64_t *)bright, (const uint64_t *)raw, n, H, W, (uint64_t *)raw));   // :566
    }
    {
        ipx_range r("compute_persistent_mask");
        // (no keep list: the raw planes of discarded frames are all zero here, which resets the runs just the same)
        IPX_TRY(ipx_persistent_mask(ctx, (const uint64_t *)raw, nullptr, n, H, W, (uint64_t *)pers));    // :575
    }
    {
        ipx_range r("filter_consecutive_pixels+compute_static_mask");
        IPX_TRY(ipx_static_maps(ctx, frames, (const uint64_t *)raw, (const uint64_t *)pers, nullptr, n, H, W, heat, valid));   // :580-590
        IPX_TRY(ipx_static_from_maps(ctx, heat, valid, H, W, (uint64_t *)stat));
    }
    {
        ipx_range r("apply_static_suppression+bright_area_counters");
        IPX_TRY(ipx_finalize(ctx, frames, (const uint64_t *)raw, (const uint64_t *)pers, (const uint64_t *)stat, keep, n, H, W,
                             (uint64_t *)fin, fin3));                                             // :592-605
    }
    {
        ipx_range r("compute_cluster_stats");
        IPX_TRY(ipx_cluster_stats(ctx, frames, (const uint64_t *)fin, keep, n, H, W, pow2_table, tdim, cluster, nullptr));   // :608
    }
    return ipx_merge_counts(ctx, fin3, dense2, n, counts);
}

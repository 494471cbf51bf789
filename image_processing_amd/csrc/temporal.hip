// temporal.hip -- chunk-global temporal maps and the per-frame integer counters (gfx950).
//
//   compute_persistent_mask    ip.py:270-289   longest run of consecutive flagged frames >= T
//   filter_consecutive_pixels  ip.py:292-310   clean = raw & ~persistent
//   compute_static_mask        ip.py:438-465   heat / valid percentage map > static_threshold
//   apply_static_suppression   ip.py:468-481   final = clean & ~(persistent | static)
//   estimate_... / find_bright_area_estimates / _get_final_count  ip.py:770-847, 351-365
//     (their per-frame integer counters; the float glue stays on the host)
//
// All masks are packed bit-planes; a lane owns one 64-pixel word and streams the n planes.
#include "ipx_internal.h"

namespace {

// Bit-sliced saturating run-length counters: K planes of 64 lanes each.  The n planes are cut into groups of
// `group` planes (blockIdx.y); a group also walks the T-1 planes before its own, so every run of >= T planes is
// seen whole by the group its T-th plane falls into (the same argument as for frame-range shards, include/ipx.h),
// and the groups' results are OR-ed into `out` (zeroed by the caller).  A lane that walked all n planes alone
// was one chain of n dependent-issue loads on half the CUs.
template <int K, bool HAS_KEEP>
__global__ void persistent_kernel(const uint64_t *raw, const uint8_t *keep, int n, size_t words, int T, int group,
                                  uint64_t *out)
{
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    const int i1 = min(n, ((int)blockIdx.y + 1) * group);
    const int i0 = max(0, (int)blockIdx.y * group - (T - 1));
    uint64_t c[K], tb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { c[k] = 0; tb[k] = ((T >> k) & 1) ? ~0ull : 0ull; }
    uint64_t pers = 0;
    constexpr int B = 8;                                  // planes loaded together: B independent loads in flight
#pragma unroll 1
    for (int i = i0; i < i1; i += B) {
        uint64_t mm[B];
#pragma unroll
        for (int u = 0; u < B; ++u) {
            // planes past the end are read as the last plane and zeroed (no branch between the loads); a zero plane
            // after the last one only resets counters that are not looked at again
            const int iu = min(i + u, i1 - 1);
            uint64_t m = raw[(size_t)iu * words + w];
            if (HAS_KEEP) m &= keep[iu] ? ~0ull : 0ull;   // None resets runs
            mm[u] = i + u < i1 ? m : 0ull;
        }
#pragma unroll
        for (int u = 0; u < B; ++u) {
            const uint64_t m = mm[u];
            // cnt = m ? min(cnt + 1, 2^K - 1) : 0
            uint64_t carry = m;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const uint64_t t = c[k] & carry;
                c[k] ^= carry;
                carry = t;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) c[k] = (c[k] | carry) & m;     // saturate on overflow, reset where m = 0
            // ge = cnt >= T, MSB-first comparison against T (tb[k] = all ones where bit k of T is set; branch-free)
            uint64_t gt = 0, eq = ~0ull;
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                gt |= eq & c[k] & ~tb[k];
                eq &= ~(c[k] ^ tb[k]);
            }
            pers |= gt | eq;
        }
    }
    if (pers) atomicOr(reinterpret_cast<unsigned long long *>(out + w), (unsigned long long)pers);
}

__global__ void fill_plane_kernel(uint64_t *out, int H, int W, int wpr)
{
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= (size_t)H * wpr) return;
    const int k = (int)(w % wpr);
    const int valid = W - 64 * k;
    out[w] = valid >= 64 ? ~0ull : (1ull << valid) - 1;
}

// heat = sum clean, valid = sum !(frame > bt & !clean); one lane owns 4 consecutive pixels of a group of
// SM_GROUP frames (blockIdx.z) and adds its partial counts to the maps (zeroed by the caller; integer adds,
// so the order does not matter).  A lane that walked all n frames alone was bound by load latency.
constexpr int SM_GROUP = 64;
__global__ void static_maps_kernel(const uint8_t *frames, const uint64_t *raw, const uint64_t *pers,
                                   const uint8_t *keep, int n, int H, int W, int wpr, int frame_gt,
                                   int32_t *heat, int32_t *valid)
{
    const int xq = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y;
    if (xq >= W) return;
    const int i0 = blockIdx.z * SM_GROUP, i1 = min(n, i0 + SM_GROUP);
    const size_t words = (size_t)H * wpr;
    const size_t wi = (size_t)y * wpr + (xq >> 6);
    const int sh = xq & 63;
    const uint64_t notp = ~pers[wi];
    const bool aligned = (W & 3) == 0;
    const size_t P = (size_t)H * W;
    const uint8_t *px = frames + (size_t)y * W + xq;
    int hc[4] = {0, 0, 0, 0}, vc[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int i = i0; i < i1; ++i) {
        uint32_t f;
        if (aligned) f = *reinterpret_cast<const uint32_t *>(px + (size_t)i * P);
        else {
            f = 0;
            for (int j = 0; j < 4; ++j)
                if (xq + j < W) f |= (uint32_t)px[(size_t)i * P + j] << (8 * j);
        }
        const uint32_t m = (keep == nullptr || keep[i]) ? (uint32_t)((raw[(size_t)i * words + wi] & notp) >> sh) & 0xf : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int mj = (m >> j) & 1;
            const int bright = ((int)((f >> (8 * j)) & 0xff) > frame_gt) & (mj ^ 1);
            hc[j] += mj;
            vc[j] += bright ^ 1;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (xq + j < W) {
            if (hc[j]) atomicAdd(&heat[(size_t)y * W + xq + j], hc[j]);
            if (vc[j]) atomicAdd(&valid[(size_t)y * W + xq + j], vc[j]);
        }
}

// The same sums for images whose rows are multiples of 16 pixels: a thread owns 16 consecutive pixels (one 16-byte
// load per frame, a wave reads 1 KiB of a row) of a group of SMV_GROUP <= 255 frames and keeps, per pixel, two byte
// counters in SWAR form: heat (frames with the clean bit set) and INVALID (frame > bt and not clean).  Both are
// sparse, so few lanes reach the atomics at the end: valid starts at n (filled by the caller) and loses the
// invalid counts.
constexpr int SMV_GROUP = 128;
template <bool HAS_KEEP>
__global__ __launch_bounds__(256) void static_maps16_kernel(const uint8_t *frames, const uint64_t *raw, const uint64_t *pers,
                                                            const uint8_t *keep, int n, int H, int W, int wpr, int frame_gt,
                                                            int32_t *heat, int32_t *valid)
{
    const int cpr = W >> 4;                                  // 16-pixel chunks per row
    const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= (size_t)H * cpr) return;
    const int y = (int)(chunk / cpr), xq = (int)(chunk - (size_t)y * cpr) << 4;
    const int i0 = blockIdx.y * SMV_GROUP, i1 = min(n, i0 + SMV_GROUP);
    const size_t words = (size_t)H * wpr, P = (size_t)H * W;
    const size_t wi = (size_t)y * wpr + (xq >> 6);
    const int sh = xq & 63;
    const uint64_t notp = ~pers[wi];
    const uint8_t *px = frames + (size_t)y * W + xq;
    // x > frame_gt per byte, bit 7 of the byte (see swar in detect.hip): mode by the threshold's top bit
    const int T = frame_gt;
    const uint32_t add = (uint32_t)(0x7f - (T & 0x7f)) * 0x01010101u;
    const bool none = T >= 255, all = T < 0, low = T < 128;
    uint32_t hacc[4] = {0, 0, 0, 0}, iacc[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int i = i0; i < i1; ++i) {
        const uint4 f = *reinterpret_cast<const uint4 *>(px + (size_t)i * P);
        uint32_t m16 = (uint32_t)((raw[(size_t)i * words + wi] & notp) >> sh) & 0xffffu;
        if (HAS_KEEP) m16 = keep[i] ? m16 : 0u;
        const uint32_t d[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t lowsum = (d[q] & 0x7f7f7f7fu) + add;
            uint32_t g = none ? 0u : (all ? 0x80808080u : ((low ? (lowsum | d[q]) : (lowsum & d[q])) & 0x80808080u));
            g >>= 7;                                                       // 0x01 per byte with frame > bt
            const uint32_t ms = (((m16 >> (4 * q)) & 0xfu) * 0x00204081u) & 0x01010101u;   // clean bits, one per byte
            hacc[q] += ms;
            iacc[q] += g & ~ms;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hc = (hacc[q] >> (8 * j)) & 0xff, ic = (iacc[q] >> (8 * j)) & 0xff;
            const size_t o = (size_t)y * W + xq + 4 * q + j;
            if (hc) atomicAdd(&heat[o], hc);
            if (ic) atomicSub(&valid[o], ic);
        }
}

__global__ void fill_i32_kernel(int32_t *p, size_t count, int32_t v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

__global__ void or_planes_kernel(const uint64_t *planes, int count, size_t words, uint64_t *out)
{
    const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= words) return;
    uint64_t r = 0;
    for (int i = 0; i < count; ++i) r |= planes[(size_t)i * words + w];
    out[w] = r;
}

// percentage_map > static_threshold where valid > min_valid_frames, ip.py:461-465 (float64)
__global__ void static_from_maps_kernel(const int32_t *heat, const int32_t *valid, int H, int W, int wpr,
                                        int min_valid, double thr, uint64_t *out)
{
    const int lane = threadIdx.x & 63;
    const size_t word = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (word >= (size_t)H * wpr) return;
    const size_t y = word / wpr;
    const int x = (int)(word - y * wpr) * 64 + lane;
    bool on = false;
    if (x < W) {
        const int v = valid[y * W + x];
        if (v > min_valid) {
            double pct = (double)heat[y * W + x] / (double)v;
            pct = pct * 100.0;
            on = pct > thr;
        }
    }
    const unsigned long long b = __ballot(on);
    if (lane == 0) out[word] = b;
}

// final planes + per-frame counters {sum(final), #(final & f < est_lt), #(final & f > frame_gt)}.
// A thread owns two adjacent words of one frame; the counters are reduced per wave, then per workgroup in LDS,
// and leave as at most three atomics per workgroup (one per wave and counter saturated the atomic unit: the
// planes are sparse, but nearly every wave holds a few set bits).
constexpr int FIN_WPT = 2;
__global__ __launch_bounds__(256) void finalize_kernel(const uint8_t *frames, const uint64_t *raw, const uint64_t *pers,
                                                       const uint64_t *stat, const uint8_t *keep, int H, int W, int wpr,
                                                       int est_lt, int frame_gt, uint64_t *fin, unsigned long long *counts)
{
    __shared__ unsigned s_c[3];
    const size_t words = (size_t)H * wpr;
    const int i = blockIdx.y;
    const size_t w0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * FIN_WPT;
    if (threadIdx.x < 3) s_c[threadIdx.x] = 0;
    __syncthreads();
    const bool kept = keep == nullptr || keep[i];
    unsigned c0 = 0, c1 = 0, c2 = 0;
    uint64_t r[FIN_WPT], bad[FIN_WPT];
#pragma unroll
    for (int k = 0; k < FIN_WPT; ++k) {
        const size_t w = w0 + k;
        r[k] = (kept && w < words) ? raw[(size_t)i * words + w] : 0ull;
        bad[k] = w < words ? (pers[w] | (stat ? stat[w] : 0ull)) : 0ull;
    }
#pragma unroll
    for (int k = 0; k < FIN_WPT; ++k) {
        const size_t w = w0 + k;
        const uint64_t m = r[k] & ~bad[k];
        if (fin && w < words) fin[(size_t)i * words + w] = m;
        if (m) {
            c0 += __builtin_popcountll(m);
            const size_t y = w / wpr;
            const uint8_t *row = frames + ((size_t)i * H + y) * W + (w - y * wpr) * 64;
            uint64_t rem = m;
            while (rem) {
                const int b = __builtin_ctzll(rem);
                rem &= rem - 1;
                const int f = row[b];
                c1 += f < est_lt;
                c2 += f > frame_gt;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off);
        c1 += __shfl_down(c1, off);
        c2 += __shfl_down(c2, off);
    }
    if ((threadIdx.x & 63) == 0 && (c0 | c1 | c2)) {
        atomicAdd(&s_c[0], c0); atomicAdd(&s_c[1], c1); atomicAdd(&s_c[2], c2);
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_c[threadIdx.x]) atomicAdd(counts + (size_t)i * 3 + threadIdx.x, (unsigned long long)s_c[threadIdx.x]);
}

// per-frame {#(frame < est_lt), #(frame > frame_gt)}: the mask-independent inputs of ip.py:800-805, :840
__global__ __launch_bounds__(256) void dense_counts_kernel(const uint8_t *frames, size_t P, int est_lt, int frame_gt,
                                                          unsigned long long *counts)
{
    const uint8_t *f = frames + (size_t)blockIdx.y * P;
    unsigned c0 = 0, c1 = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (size_t)gridDim.x * 256) {
        const int v = f[i];
        c0 += v < est_lt;
        c1 += v > frame_gt;
    }
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off);
        c1 += __shfl_down(c1, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(counts + (size_t)blockIdx.y * 2 + 0, (unsigned long long)c0);
        atomicAdd(counts + (size_t)blockIdx.y * 2 + 1, (unsigned long long)c1);
    }
}

}  // namespace

extern "C" int ipx_dense_counts(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W, int64_t *counts)
{
    IPX_REQUIRE(ctx && frames && counts, "null pointer");
    IPX_REQUIRE(n > 0 && n <= 65535 && H > 0 && W > 0, "bad sizes (n <= 65535 frames per call)");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const ipx_derived d = ipx_derive(ctx->p);
    const size_t P = (size_t)H * W;
    IPX_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * 2 * (size_t)n, ctx->stream));
    const unsigned bx = (unsigned)((P + 256 * 16 - 1) / (256 * 16));
    IPX_LAUNCH(ctx, KID_MISC, (double)n * P, dense_counts_kernel, dim3(bx < 1 ? 1 : bx, n), dim3(256), frames, P, d.est_lt,
               d.frame_gt, (unsigned long long *)counts);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_persistent_mask(ipx_ctx *ctx, const uint64_t *raw_bits, const uint8_t *keep, int n, int H, int W,
                                   uint64_t *persistent_bits)
{
    IPX_REQUIRE(ctx && raw_bits && persistent_bits, "null pointer");
    IPX_REQUIRE(n > 0 && H > 0 && W > 0, "bad sizes");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const int wpr = ipx_wpr(W);
    const size_t words = (size_t)H * wpr;
    const dim3 grid((unsigned)((words + 255) / 256)), block(256);
    const int T = ctx->p.consecutive_threshold;
    if (T <= 0) {                     // longest >= T holds for every pixel, ip.py:289
        IPX_LAUNCH(ctx, KID_PERSISTENT, words * 8, fill_plane_kernel, grid, block, persistent_bits, H, W, wpr);
    } else if (T > n) {               // no run can be that long
        IPX_CHECK_HIP(hipMemsetAsync(persistent_bits, 0, words * 8, ctx->stream));
    } else {
        // groups of >= 32 planes, and at least 4 (T-1) so that the overlap stays below a quarter of the reads
        long long group = 32;
        if (group < 4LL * (T - 1)) group = 4LL * (T - 1);
        if (group > n) group = n;
        const unsigned ng = (unsigned)((n + group - 1) / group);
        const dim3 g2(grid.x, ng);
        IPX_CHECK_HIP(hipMemsetAsync(persistent_bits, 0, words * 8, ctx->stream));
        const double ab = (n + 1.0) * words * 8;
#define IPX_PERS(K)                                                                                                        \
        do {                                                                                                               \
            if (keep) IPX_LAUNCH(ctx, KID_PERSISTENT, ab, (persistent_kernel<K, true>), g2, block, raw_bits, keep, n, words, T, (int)group, persistent_bits); \
            else IPX_LAUNCH(ctx, KID_PERSISTENT, ab, (persistent_kernel<K, false>), g2, block, raw_bits, keep, n, words, T, (int)group, persistent_bits);     \
        } while (0)
        if (T < 4) IPX_PERS(2);
        else if (T < 256) IPX_PERS(8);
        else IPX_PERS(24);
#undef IPX_PERS
    }
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_static_maps(ipx_ctx *ctx, const uint8_t *frames, const uint64_t *raw_bits,
                               const uint64_t *persistent_bits, const uint8_t *keep, int n, int H, int W,
                               int32_t *heat, int32_t *valid)
{
    IPX_REQUIRE(ctx && frames && raw_bits && persistent_bits && heat && valid, "null pointer");
    IPX_REQUIRE(n > 0 && H > 0 && W > 0, "bad sizes");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const ipx_derived d = ipx_derive(ctx->p);
    const double ab = (double)n * ((double)H * W + (double)H * ipx_wpr(W) * 8) + 8.0 * H * W + (double)H * ipx_wpr(W) * 8;
    if ((W & 15) == 0 && (reinterpret_cast<uintptr_t>(frames) & 15) == 0) {
        const size_t P = (size_t)H * W, chunks = P / 16;
        IPX_CHECK_HIP(hipMemsetAsync(heat, 0, sizeof(int32_t) * P, ctx->stream));
        IPX_LAUNCH(ctx, KID_MISC, 4.0 * P, fill_i32_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), valid, P, (int32_t)n);
        const dim3 g16((unsigned)((chunks + 255) / 256), (n + SMV_GROUP - 1) / SMV_GROUP), b16(256);
        if (keep) IPX_LAUNCH(ctx, KID_STATIC_MAPS, ab, static_maps16_kernel<true>, g16, b16, frames, raw_bits, persistent_bits, keep, n, H, W,
                             ipx_wpr(W), d.frame_gt, heat, valid);
        else IPX_LAUNCH(ctx, KID_STATIC_MAPS, ab, static_maps16_kernel<false>, g16, b16, frames, raw_bits, persistent_bits, keep, n, H, W,
                        ipx_wpr(W), d.frame_gt, heat, valid);
        IPX_LAUNCH_CHECK();
        return IPX_OK;
    }
    const int quads = (W + 3) / 4;
    const dim3 grid((quads + 63) / 64, H, (n + SM_GROUP - 1) / SM_GROUP), block(64);
    IPX_CHECK_HIP(hipMemsetAsync(heat, 0, sizeof(int32_t) * (size_t)H * W, ctx->stream));
    IPX_CHECK_HIP(hipMemsetAsync(valid, 0, sizeof(int32_t) * (size_t)H * W, ctx->stream));
    IPX_LAUNCH(ctx, KID_STATIC_MAPS, (double)n * ((double)H * W + (double)H * ipx_wpr(W) * 8) + 8.0 * H * W + (double)H * ipx_wpr(W) * 8, static_maps_kernel, grid, block, frames, raw_bits, persistent_bits, keep,
                       n, H, W, ipx_wpr(W), d.frame_gt, heat, valid);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_static_from_maps(ipx_ctx *ctx, const int32_t *heat, const int32_t *valid, int H, int W,
                                    uint64_t *static_bits)
{
    IPX_REQUIRE(ctx && heat && valid && static_bits, "null pointer");
    IPX_REQUIRE(H > 0 && W > 0, "bad sizes");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const int wpr = ipx_wpr(W);
    const size_t words = (size_t)H * wpr;
    IPX_LAUNCH(ctx, KID_STATIC_FROM_MAPS, 8.0 * H * W + words * 8, static_from_maps_kernel, dim3((unsigned)((words + 3) / 4)), dim3(256), heat, valid, H, W, wpr, ctx->p.min_valid_frames, ctx->p.static_threshold, static_bits);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_finalize(ipx_ctx *ctx, const uint8_t *frames, const uint64_t *raw_bits,
                            const uint64_t *persistent_bits, const uint64_t *static_bits, const uint8_t *keep,
                            int n, int H, int W, uint64_t *final_bits, int64_t *counts)
{
    IPX_REQUIRE(ctx && frames && raw_bits && persistent_bits && counts, "null pointer");
    IPX_REQUIRE(n > 0 && n <= 65535 && H > 0 && W > 0, "bad sizes (n <= 65535 frames per call)");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const ipx_derived d = ipx_derive(ctx->p);
    const int wpr = ipx_wpr(W);
    const size_t words = (size_t)H * wpr;
    IPX_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int64_t) * 3 * (size_t)n, ctx->stream));
    const dim3 grid((unsigned)((words + 256 * FIN_WPT - 1) / (256 * FIN_WPT)), n), block(256);
    IPX_LAUNCH(ctx, KID_FINALIZE, (2.0 * n + 2.0) * words * 8, finalize_kernel, grid, block, frames, raw_bits, persistent_bits, static_bits,
                       keep, H, W, wpr, d.est_lt, d.frame_gt, final_bits, (unsigned long long *)counts);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_or_planes(ipx_ctx *ctx, const uint64_t *planes, int count, int H, int W, uint64_t *out)
{
    IPX_REQUIRE(ctx && planes && out, "null pointer");
    IPX_REQUIRE(count > 0 && H > 0 && W > 0, "bad sizes");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const size_t words = (size_t)H * ipx_wpr(W);
    IPX_LAUNCH(ctx, KID_MISC, (count + 1.0) * words * 8, or_planes_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), planes, count,
                       words, out);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

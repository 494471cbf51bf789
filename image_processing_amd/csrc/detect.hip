// detect.hip -- temporal background + brightness/contrast tests -> packed bit-planes (gfx950).
//
// Replaces, for a resident chunk of uint8 frames:
//   compute_background/_find_background  ip.py:183-252  (numba typing: SURVEY App. A.1)
//   get_damaged_pixel_mask               ip.py:635-648  (f32 threshold chain A.2, 30x30 box mean A.3)
//   (background > brightness_threshold)  ip.py:743      (input of remove_bright_regions)
//
// One 256-thread workgroup owns a 64x64 pixel tile of one frame.  The 30x30 window sum is an
// exact integer: the (64+29)^2 window is staged in LDS, summed horizontally (u16) then
// vertically (i32).  `frame > S*(1/900.0)` in double is equivalent to `900*frame > S` in
// integers (S*(1/900.0) == v exactly when S == 900 v, and the next integers are 1/900 away).
// Mask bits leave the kernel as 64-bit words built by wave64 ballots.
//
// Floating-point contract (must stay bit-exact with the reference): no FMA contraction, IEEE
// f32 divide, IEEE f64 divide/sqrt -- the library is built with -ffp-contract=off and without
// fast-math.
#include "ipx_internal.h"

namespace {

constexpr int TW = 64, TH = 64, KS = 30, AX = 15;
constexpr int WIN_H = TH + KS - 1;       // 93 rows
constexpr int WIN_STRIDE = 100;          // bytes per staged row: 96 used (x0-16 .. x0+79), padded
constexpr int HS_STRIDE = 64;            // u16 per row

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ uint32_t load4(const uint8_t *row, int x, int W, bool aligned)
{
    // four pixels x..x+3 of a row as little-endian bytes; pixels past W read as 0
    if (aligned && x + 3 < W) return *reinterpret_cast<const uint32_t *>(row + x);
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (x + j < W) v |= (uint32_t)row[x + j] << (8 * j);
    return v;
}

// Divisions without the IEEE divide sequence, bit-exact all the same:
// * f64: for y = RN(1/b), q = RN(a*y) corrected twice by q += (a - b*q)*y (the residual is exact in an fma).
//   The first correction makes q faithful, the second is then correctly rounded (Markstein's theorem).
// * f32 quotients of a 24-bit a by an integer b <= 255: RN32(RN64(a * RN64(1/b))).  a/b is never closer
//   than 2^-32 (relative) to a midpoint between two floats unless it is a float itself, and the f64 product
//   is off by less than 2^-51: the final rounding cannot differ.  (tests/test_abi_and_host.py walks through
//   every S/n and Si/c, and a sample of bg/255.)
__device__ __forceinline__ double div_by(double a, double b, double y)
{
    double q = a * y;
    double r = __builtin_fma(-b, q, a);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}

// _find_background for one pixel from its n temporal neighbours (byte j of v[0..n)); rcp[i] = 1.0 / i.
// FULL: n == MAXN is known at compile time (every frame that is 'radius' away from both chunk ends), so the
// neighbour loops carry no predication.
template <int MAXN, bool FULL>
__device__ __forceinline__ float background_px(const uint32_t (&v)[MAXN], int n, int j, double coeff, const double *rcp)
{
    if (FULL) n = MAXN;
    const double yn = FULL ? 1.0 / (double)MAXN : rcp[n];
    int S = 0;
#pragma unroll
    for (int k = 0; k < MAXN; ++k)
        if (FULL || k < n) S += (v[k] >> (8 * j)) & 0xff;
    const float mean = (float)((double)S * yn);  // == (float)S / (float)n == f32(f64(S)/n)
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < MAXN; ++k)
        if (FULL || k < n) {
            const float diff = (float)((v[k] >> (8 * j)) & 0xff) - mean;
            const float d2 = diff * diff;
            acc += (double)d2;
        }
    const float sd = (float)sqrt(div_by(acc, (double)n, yn));
    const double thr = (double)mean + coeff * (double)sd;
    // (double)b <= thr for a byte b  <=>  b <= T  (a NaN threshold admits nothing, as in the reference)
    const int T = thr >= 0.0 ? (int)fmin(thr, 255.0) : -1;
    int Si = 0, c = 0;
#pragma unroll
    for (int k = 0; k < MAXN; ++k)
        if (FULL || k < n) {
            const int b = (v[k] >> (8 * j)) & 0xff;
            if (b <= T) { Si += b; ++c; }
        }
    return c > 0 ? (float)((double)Si * rcp[c]) : mean;
}

__device__ __forceinline__ float damage_threshold_f32(float bg)
{
    float t = (float)((double)bg * (1.0 / 255.0));   // == bg / 255.0f; three separately rounded f32 ops, ip.py:638
    t = t * 225.0f;
    t = 30.0f + t;
    return fmaxf(30.0f, t);
}

struct DetectArgs {
    const uint8_t *frames;   // [n][H][W]
    const uint8_t *keep;     // [n] or null (= all kept)
    uint64_t *raw_bits;      // [count][H*wpr]
    uint64_t *bright_bits;   // [count][H*wpr]
    unsigned long long *dense;  // [count][2] or null
    float *bg_out;           // optional (H,W) f32 for the stage entry point (count == 1)
    int n, H, W, first, radius;
    int frame_gt, est_lt;
    float bg_gt;
    double coeff;
};

template <int MAXN>
__global__ __launch_bounds__(256) void detect_kernel(DetectArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[WIN_H * WIN_STRIDE];
    __shared__ __attribute__((aligned(16))) uint16_t s_hs[WIN_H * HS_STRIDE];   // later: flag bytes
    __shared__ __attribute__((aligned(16))) int32_t s_vs[TH * TW];
    __shared__ unsigned int s_cnt[2];
    __shared__ double s_rcp[MAXN + 1];

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const int wpr = (W + 63) >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const int fz = blockIdx.z;              // plane within this call
    const int fi = a.first + fz;            // frame within the chunk
    uint64_t *rawp = a.raw_bits + (size_t)fz * H * wpr;
    uint64_t *brp = a.bright_bits ? a.bright_bits + (size_t)fz * H * wpr : nullptr;

    if (a.keep && !a.keep[fi]) {            // discarded frame: zero planes (wave-uniform exit)
        if (tid < TH && y0 + tid < H) {
            rawp[(size_t)(y0 + tid) * wpr + blockIdx.x] = 0;
            if (brp) brp[(size_t)(y0 + tid) * wpr + blockIdx.x] = 0;
        }
        return;
    }
    if (tid < 2) s_cnt[tid] = 0;
    if (tid <= MAXN) s_rcp[tid] = tid ? 1.0 / (double)tid : 0.0;

    const uint8_t *frame = a.frames + (size_t)fi * H * W;
    const bool aligned = (W & 3) == 0;

    // ---- phase A: stage the reflected window, columns x0-16 .. x0+79 ------------------------
    const bool interior_x = aligned && x0 - 16 >= 0 && x0 + 80 <= W;
    for (int idx = tid; idx < WIN_H * 24; idx += 256) {
        const int r = idx / 24, q = idx - r * 24;
        const int gy = reflect101(y0 - AX + r, H);
        const uint8_t *row = frame + (size_t)gy * W;
        uint32_t v;
        if (interior_x) {
            v = *reinterpret_cast<const uint32_t *>(row + x0 - 16 + 4 * q);
        } else {
            v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v |= (uint32_t)row[reflect101(x0 - 16 + 4 * q + j, W)] << (8 * j);
        }
        *reinterpret_cast<uint32_t *>(s_win + r * WIN_STRIDE + 4 * q) = v;
    }
    __syncthreads();

    // ---- phase B: horizontal 30-sums, hs[r][x] = sum_{i<30} win[r][x+1+i] ---------------------
    for (int item = tid; item < WIN_H * 4; item += 256) {
        const int r = item >> 2, seg = item & 3;
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(s_win + r * WIN_STRIDE) + seg * 4;
        uint32_t d[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) d[j] = rp[j];
        auto byte_at = [&](int k) -> int { return (d[k >> 2] >> (8 * (k & 3))) & 0xff; };
        int s = 0;
#pragma unroll
        for (int i = 1; i <= 30; ++i) s += byte_at(i);
        uint32_t out[8];
        int prev = s;
#pragma unroll
        for (int t = 1; t < 16; ++t) {
            s += byte_at(30 + t) - byte_at(t);
            if (t & 1) out[t >> 1] = (uint32_t)prev | ((uint32_t)s << 16);
            else prev = s;
        }
        uint32_t *hp = reinterpret_cast<uint32_t *>(s_hs + r * HS_STRIDE + seg * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[j] = out[j];
    }
    __syncthreads();

    // ---- phase C: vertical 30-sums, vs[y][x] = sum_{i<30} hs[y+i][x] --------------------------
    {
        const int x = tid & 63, q = tid >> 6;
        int s = 0;
#pragma unroll 6
        for (int i = 0; i < 30; ++i) s += s_hs[(16 * q + i) * HS_STRIDE + x];
        s_vs[(16 * q) * TW + x] = s;
#pragma unroll 5
        for (int t = 1; t < 16; ++t) {
            s += (int)s_hs[(16 * q + 29 + t) * HS_STRIDE + x] - (int)s_hs[(16 * q + t - 1) * HS_STRIDE + x];
            s_vs[(16 * q + t) * TW + x] = s;
        }
    }
    __syncthreads();

    // ---- phase D: per-pixel tests, 4 pixels per thread, 16 rows per step ---------------------
    uint8_t *s_flag = reinterpret_cast<uint8_t *>(s_hs);   // [2][TH][TW] bytes: damaged, bright
    const int lo = max(0, fi - a.radius), hi = min(a.n, fi + a.radius + 1);
    const int nn = hi - lo - 1;                              // neighbours (centre excluded)
    const int xq = (tid & 15) * 4;
    const bool full = nn == MAXN;                            // workgroup-uniform
    unsigned int c_lt = 0, c_gt = 0;
#pragma unroll 1
    for (int step = 0; step < 4; ++step) {
        const int ty = (tid >> 4) + 16 * step;
        const int gy = y0 + ty, gx = x0 + xq;
        uint32_t dm = 0, bm = 0;
        if (gy < H && gx < W) {
            uint32_t v[MAXN];
            const int cpos = fi - lo;                       // centre's slot in the window
#pragma unroll
            for (int s = 0; s < MAXN; ++s) {
                const int f = lo + s + (s >= cpos ? 1 : 0);
                v[s] = s < nn ? load4(a.frames + ((size_t)f * H + gy) * W, gx, W, aligned) : 0u;
            }
            const uint32_t cen = *reinterpret_cast<const uint32_t *>(s_win + (ty + AX) * WIN_STRIDE + 16 + xq);
            const int4 vs4 = *reinterpret_cast<const int4 *>(s_vs + ty * TW + xq);
            const int vs[4] = {vs4.x, vs4.y, vs4.z, vs4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (gx + j < W) {
                    const int pix = (cen >> (8 * j)) & 0xff;
                    const float bg = full ? background_px<MAXN, true>(v, nn, j, a.coeff, s_rcp) : background_px<MAXN, false>(v, nn, j, a.coeff, s_rcp);
                    if (a.bg_out) a.bg_out[(size_t)gy * W + gx + j] = bg;
                    const float thr = damage_threshold_f32(bg);
                    const bool dmg = ((float)pix > thr) && (900 * pix > vs[j]);
                    dm |= (uint32_t)dmg << (8 * j);
                    bm |= (uint32_t)(bg > a.bg_gt) << (8 * j);
                    c_lt += pix < a.est_lt;
                    c_gt += pix > a.frame_gt;
                }
            }
        }
        *reinterpret_cast<uint32_t *>(s_flag + ty * TW + xq) = dm;
        *reinterpret_cast<uint32_t *>(s_flag + TH * TW + ty * TW + xq) = bm;
    }
    if (a.dense) {
        // wave-level reduce, then one LDS atomic per wave
        for (int off = 32; off > 0; off >>= 1) {
            c_lt += __shfl_down(c_lt, off);
            c_gt += __shfl_down(c_gt, off);
        }
        if ((tid & 63) == 0) { atomicAdd(&s_cnt[0], c_lt); atomicAdd(&s_cnt[1], c_gt); }
    }
    __syncthreads();

    // ---- phase E: ballot-pack 64 rows x 64 columns into one word per row ----------------------
    {
        const int lane = tid & 63, wv = tid >> 6;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int ty = wv * 16 + i;
            const unsigned long long wd = __ballot(s_flag[ty * TW + lane] != 0);
            const unsigned long long wb = __ballot(s_flag[TH * TW + ty * TW + lane] != 0);
            if (lane == 0 && y0 + ty < H) {
                rawp[(size_t)(y0 + ty) * wpr + blockIdx.x] = wd;
                if (brp) brp[(size_t)(y0 + ty) * wpr + blockIdx.x] = wb;
            }
        }
    }
    if (a.dense && tid < 2) atomicAdd(&a.dense[(size_t)fz * 2 + tid], (unsigned long long)s_cnt[tid]);
}

// get_damaged_pixel_mask as a stage (caller supplies the background, f32 or f64), ip.py:635-648.
template <typename T>
__global__ void damaged_mask_kernel(const uint8_t *frame, int H, int W, const T *bg, uint8_t *mask, T *thr_out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    int S = 0;
    for (int dy = 0; dy < KS; ++dy) {
        const uint8_t *row = frame + (size_t)reflect101(y - AX + dy, H) * W;
        for (int dx = 0; dx < KS; ++dx) S += row[reflect101(x - AX + dx, W)];
    }
    const T b = bg[(size_t)y * W + x];
    T t = b / (T)255;
    t = t * (T)225;
    t = (T)30 + t;
    t = t > (T)30 ? t : (t != t ? t : (T)30);      // np.maximum(30, t) propagates NaN
    const int pix = frame[(size_t)y * W + x];
    if (thr_out) thr_out[(size_t)y * W + x] = t;
    mask[(size_t)y * W + x] = ((T)pix > t) && (900 * pix > S);
}

}  // namespace

static int launch_detect(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W, int first, int count,
                         const uint8_t *keep, uint64_t *raw_bits, uint64_t *bright_bits,
                         int64_t *dense_counts, float *bg_out)
{
    const ipx_derived d = ipx_derive(ctx->p);
    DetectArgs a;
    a.frames = frames; a.keep = keep; a.raw_bits = raw_bits; a.bright_bits = bright_bits;
    a.dense = reinterpret_cast<unsigned long long *>(dense_counts); a.bg_out = bg_out;
    a.n = n; a.H = H; a.W = W; a.first = first; a.radius = ctx->p.sliding_window_radius;
    a.frame_gt = d.frame_gt; a.est_lt = d.est_lt; a.bg_gt = d.bg_gt; a.coeff = ctx->p.pixel_std_coeff;
    if (dense_counts)
        IPX_CHECK_HIP(hipMemsetAsync(dense_counts, 0, sizeof(int64_t) * 2 * (size_t)count, ctx->stream));
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, count), block(256);
    const int r = ctx->p.sliding_window_radius;
    // algorithmic bytes: centre frame + its temporal neighbours read once, two bit-planes written
    double ab = 2.0 * count * H * ipx_wpr(W) * 8;
    for (int i = first; i < first + count; ++i) {
        const int lo = i - r < 0 ? 0 : i - r, hi = i + r + 1 > n ? n : i + r + 1;
        ab += (double)(hi - lo) * H * W;
    }
    if (r <= 3) IPX_LAUNCH(ctx, KID_DETECT, ab, detect_kernel<6>, grid, block, a);
    else if (r <= 8) IPX_LAUNCH(ctx, KID_DETECT, ab, detect_kernel<16>, grid, block, a);
    else {
        ipx_set_error("sliding_window_radius %d > 8 is not supported by the HIP path", r);
        return IPX_ERR_UNSUPPORTED;
    }
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_detect_raw(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W, int first, int count,
                              const uint8_t *keep, uint64_t *raw_bits, uint64_t *bright_bits,
                              int64_t *dense_counts)
{
    IPX_REQUIRE(ctx && frames && raw_bits, "null pointer");
    IPX_REQUIRE(n > 0 && H > 0 && W > 0 && first >= 0 && count > 0 && first + count <= n, "bad sizes");
    IPX_REQUIRE(ctx->p.sliding_window_radius >= 1, "sliding_window_radius must be >= 1");
    IPX_REQUIRE(n >= 2, "a chunk needs at least 2 frames (the background window excludes the centre frame)");
    if (int e = ipx_ctx_enter(ctx)) return e;
    return launch_detect(ctx, frames, n, H, W, first, count, keep, raw_bits, bright_bits, dense_counts, nullptr);
}

extern "C" int ipx_background(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W, int index, float *bg)
{
    IPX_REQUIRE(ctx && frames && bg, "null pointer");
    IPX_REQUIRE(n >= 2 && H > 0 && W > 0 && index >= 0 && index < n, "bad sizes");
    IPX_REQUIRE(ctx->p.sliding_window_radius >= 1, "sliding_window_radius must be >= 1");
    if (int e = ipx_ctx_enter(ctx)) return e;
    void *tmp;
    const size_t plane = sizeof(uint64_t) * (size_t)H * ipx_wpr(W);
    if (int e = ipx_scratch(ctx, SLOT_STAGE, plane, &tmp)) return e;
    return launch_detect(ctx, frames, n, H, W, index, 1, nullptr, (uint64_t *)tmp, nullptr, nullptr, bg);
}

extern "C" int ipx_damaged_mask(ipx_ctx *ctx, const uint8_t *frame, int H, int W, const void *bg,
                                int bg_is_f64, uint8_t *mask, void *thr)
{
    IPX_REQUIRE(ctx && frame && bg && mask, "null pointer");
    IPX_REQUIRE(H > 0 && W > 0, "bad sizes");
    if (int e = ipx_ctx_enter(ctx)) return e;
    dim3 grid((W + 63) / 64, H), block(64);
    if (bg_is_f64)
        IPX_LAUNCH(ctx, KID_DAMAGED_MASK, (double)H * W * (bg_is_f64 ? 18 : 10), damaged_mask_kernel<double>, grid, block, frame, H, W,
                           (const double *)bg, mask, (double *)thr);
    else
        IPX_LAUNCH(ctx, KID_DAMAGED_MASK, (double)H * W * (bg_is_f64 ? 18 : 10), damaged_mask_kernel<float>, grid, block, frame, H, W,
                           (const float *)bg, mask, (float *)thr);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

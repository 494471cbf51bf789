// ccl.hip -- 8-connected component labelling on packed bit-planes, and the two consumers of it:
//   remove_bright_regions           ip.py:731-767  (2x connectedComponentsWithStats + overlap test)
//   filter_damaged_pixel_clusters   ip.py:651-728  (close, CCL, size + circularity filter, stats)
//
// Labelling unit = a SEGMENT: a maximal run of set bits inside one 64-bit word.  Its id is the
// padded pixel index of its first bit, id = (word_index << 6) | bit.  One lane owns one word:
// it links each of its segments to the previous word's last segment (when contiguous) and to
// the overlapping segments of the three words above (8-connectivity), with a lock-free
// union-find (atomicMin on the parent array; the smaller id wins, so a component's root is its
// raster-first pixel -- which is also where OpenCV starts the outer contour).  Every consumer
// of the reference is invariant to label numbering, so any correct partition
// is bit-exact at the API.
#include "ipx_internal.h"

namespace {

struct Plane {
    int H, W, wpr;
    size_t words;      // H * wpr
    size_t ids;        // words * 64 (parent-array entries per plane)
};

__device__ __forceinline__ int ld_relaxed(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int find_root(const int *par, int x)
{
    int p;
    while ((p = ld_relaxed(par + x)) != x) x = p;
    return x;
}

__device__ __forceinline__ void unite(int *par, int a, int b)
{
    for (;;) {
        a = find_root(par, a);
        b = find_root(par, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(par + a, b);     // link the larger root under the smaller
        if (old == a) return;
        a = old;                                     // `a` had been linked meanwhile: carry on
    }
}

// lowest segment of `rem` (rem != 0): start bit, length, mask
__device__ __forceinline__ void first_segment(uint64_t rem, int &s, int &len, uint64_t &mask)
{
    s = __builtin_ctzll(rem);
    const uint64_t inv = ~(rem >> s);               // zeros where the run continues
    len = inv ? __builtin_ctzll(inv) : 64;
    if (len > 64 - s) len = 64 - s;
    mask = (len == 64 ? ~0ull : ((1ull << len) - 1)) << s;
}

// start bit of the segment of word u that contains bit b (u has bit b set)
__device__ __forceinline__ int segment_start(uint64_t u, int b)
{
    const uint64_t z = ~u & ((1ull << b) - 1);      // zeros below b
    return z ? 64 - __builtin_clzll(z) : 0;
}

// start bit of the last segment of u (u has bit 63 set)
__device__ __forceinline__ int last_segment_start(uint64_t u)
{
    const uint64_t z = ~u;
    return z ? 64 - __builtin_clzll(z) : 0;
}

// ---- phase 1: every segment its own root; clear the per-root accumulators -----------------------
__global__ void ccl_init_kernel(const uint64_t *bits, Plane pl, int count, int *par, int *aux0, unsigned *aux1)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    if (!rem) return;
    const size_t plane = g / pl.words;
    const int w = (int)(g - plane * pl.words);
    int *pp = par + plane * pl.ids;
    int *a0 = aux0 + plane * pl.ids;
    unsigned *a1 = aux1 ? aux1 + plane * pl.ids : nullptr;
    while (rem) {
        int s, len; uint64_t m;
        first_segment(rem, s, len, m);
        rem &= ~m;
        const int id = (w << 6) | s;
        pp[id] = id;
        a0[id] = 0;
        if (a1) a1[id] = 0;
    }
}

// ---- phase 2: link segments (left word, three words above) --------------------------------------
__global__ void ccl_merge_kernel(const uint64_t *bits, Plane pl, int count, int *par)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    const uint64_t wd = bits[g];
    if (!wd) return;
    const size_t plane = g / pl.words;
    const int w = (int)(g - plane * pl.words);
    const int y = w / pl.wpr, k = w - y * pl.wpr;
    const uint64_t *bp = bits + plane * pl.words;
    int *pp = par + plane * pl.ids;
    const uint64_t left = k > 0 ? bp[w - 1] : 0;
    uint64_t up = 0, upl = 0, upr = 0;
    if (y > 0) {
        up = bp[w - pl.wpr];
        if (k > 0) upl = bp[w - pl.wpr - 1];
        if (k + 1 < pl.wpr) upr = bp[w - pl.wpr + 1];
    }
    uint64_t rem = wd;
    while (rem) {
        int s, len; uint64_t m;
        first_segment(rem, s, len, m);
        rem &= ~m;
        const int e = s + len - 1;
        const int id = (w << 6) | s;
        if (s == 0) {
            if (left >> 63) unite(pp, id, ((w - 1) << 6) | last_segment_start(left));
            if (upl >> 63) unite(pp, id, ((w - pl.wpr - 1) << 6) | last_segment_start(upl));
        }
        if (e == 63 && (upr & 1)) unite(pp, id, (w - pl.wpr + 1) << 6);
        // bits s-1 .. e+1 of the word above
        uint64_t range = m | (m << 1) | (m >> 1);
        uint64_t t = up & range;
        while (t) {
            const int b = __builtin_ctzll(t);
            const int us = segment_start(up, b);
            int s2, len2; uint64_t m2;
            first_segment(up >> us << us, s2, len2, m2);     // the whole segment of `up` from us
            t &= ~m2;
            unite(pp, id, ((w - pl.wpr) << 6) | us);
        }
    }
}

// ---- remove_bright_regions -----------------------------------------------------------------
// bright plane: flatten + areas (aux0[root] += segment length)
__global__ void ccl_area_kernel(const uint64_t *bits, Plane pl, int count, int *par, int *aux0)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    if (!rem) return;
    const size_t plane = g / pl.words;
    const int w = (int)(g - plane * pl.words);
    int *pp = par + plane * pl.ids;
    int *a0 = aux0 + plane * pl.ids;
    while (rem) {
        int s, len; uint64_t m;
        first_segment(rem, s, len, m);
        rem &= ~m;
        const int id = (w << 6) | s;
        const int root = find_root(pp, id);
        pp[id] = root;
        atomicAdd(a0 + root, len);
    }
}

// out = segments whose component has aux0[root] >= min_area  (the `remove` plane, ip.py:750-754)
__global__ void select_big_kernel(const uint64_t *bits, Plane pl, int count, const int *par, const int *aux0,
                                  int min_area, uint64_t *out)
{
    IPX_VGPR_PAD_SAFE();
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    uint64_t r = 0;
    if (rem) {
        const size_t plane = g / pl.words;
        const int w = (int)(g - plane * pl.words);
        const int *pp = par + plane * pl.ids;
        const int *a0 = aux0 + plane * pl.ids;
        while (rem) {
            int s, len; uint64_t m;
            first_segment(rem, s, len, m);
            rem &= ~m;
            if (a0[pp[(w << 6) | s]] >= min_area) r |= m;
        }
    }
    out[g] = r;
}

// damaged plane: flatten + taint components that overlap `remove` (ip.py:761-764)
__global__ void ccl_taint_kernel(const uint64_t *bits, const uint64_t *remove, Plane pl, int count, int *par, int *aux0)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    if (!rem) return;
    const uint64_t rm = remove[g];
    const size_t plane = g / pl.words;
    const int w = (int)(g - plane * pl.words);
    int *pp = par + plane * pl.ids;
    int *a0 = aux0 + plane * pl.ids;
    while (rem) {
        int s, len; uint64_t m;
        first_segment(rem, s, len, m);
        rem &= ~m;
        const int id = (w << 6) | s;
        const int root = find_root(pp, id);
        pp[id] = root;
        if (m & rm) a0[root] = 1;
    }
}

__global__ void select_untainted_kernel(const uint64_t *bits, Plane pl, int count, const int *par, const int *aux0,
                                        uint64_t *out)
{
    IPX_VGPR_PAD_SAFE();
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    uint64_t r = 0;
    if (rem) {
        const size_t plane = g / pl.words;
        const int w = (int)(g - plane * pl.words);
        const int *pp = par + plane * pl.ids;
        const int *a0 = aux0 + plane * pl.ids;
        while (rem) {
            int s, len; uint64_t m;
            first_segment(rem, s, len, m);
            rem &= ~m;
            if (!a0[pp[(w << 6) | s]]) r |= m;
        }
    }
    out[g] = r;
}

// ---- filter_damaged_pixel_clusters ----------------------------------------------------------
// morphologyEx(MORPH_CLOSE, 3x3 cross): dilate (outside = 0) then erode (outside ignored), SURVEY App. A.4
__device__ __forceinline__ uint64_t mword(const uint64_t *bp, const Plane &pl, int y, int k)
{
    return (y >= 0 && y < pl.H && k >= 0 && k < pl.wpr) ? bp[(size_t)y * pl.wpr + k] : 0ull;
}
// One thread closes CC_RB consecutive rows of one word column: the words of its column and of the two neighbouring columns
// are loaded once for rows y0-2 .. y0+CC_RB+1 (3 (CC_RB + 4) loads for CC_RB outputs; a thread per word loaded 25 and divided
// twice).  Of the neighbouring columns' dilated words the erosion needs one bit each, which those loads determine:
//   bit 63 of dilated(y, k-1) = l63 | l62 | c0 | (row above, column k-1)63 | (row below, column k-1)63
//   bit 0  of dilated(y, k+1) = r0 | r1 | c63 | (row above, column k+1)0 | (row below, column k+1)0
// Workgroup: 32 word columns x 8 row groups; grid (columns / 32, rows / (8 CC_RB), planes).
constexpr int CC_RB = 4;
__global__ __launch_bounds__(256) void close_cross3_kernel(const uint64_t *bits, Plane pl, int count, const uint8_t *keep, uint64_t *out)
{
    const int k = (int)blockIdx.x * 32 + (int)(threadIdx.x & 31);
    const int y0 = ((int)blockIdx.y * 8 + (int)(threadIdx.x >> 5)) * CC_RB;
    const size_t plane = blockIdx.z;
    if (k >= pl.wpr || y0 >= pl.H || plane >= (size_t)count) return;
    uint64_t *op = out + plane * pl.words;
    if (keep && !keep[plane]) {
        for (int r = 0; r < CC_RB && y0 + r < pl.H; ++r) op[(size_t)(y0 + r) * pl.wpr + k] = 0;
        return;
    }
    const uint64_t *bp = bits + plane * pl.words;
    uint64_t L[CC_RB + 4], C[CC_RB + 4], R[CC_RB + 4];            // index j <-> row y0 - 2 + j
#pragma unroll
    for (int j = 0; j < CC_RB + 4; ++j) {
        const int y = y0 - 2 + j;
        L[j] = mword(bp, pl, y, k - 1); C[j] = mword(bp, pl, y, k); R[j] = mword(bp, pl, y, k + 1);
    }
    const int valid = pl.W - 64 * k;                               // clip to the image
    const uint64_t vmask = valid < 64 ? (1ull << valid) - 1 : ~0ull;
    uint64_t D[CC_RB + 2];                                          // dilated words of rows y0-1 .. y0+CC_RB (index i <-> row y0 - 1 + i)
#pragma unroll
    for (int i = 0; i < CC_RB + 2; ++i) {
        const int y = y0 - 1 + i, j = i + 1;
        const uint64_t c = C[j];
        const uint64_t d = (c | (c << 1) | (L[j] >> 63) | (c >> 1) | (R[j] << 63) | C[j - 1] | C[j + 1]) & vmask;
        D[i] = (y >= 0 && y < pl.H) ? d : 0ull;
    }
#pragma unroll
    for (int r = 0; r < CC_RB; ++r) {
        const int y = y0 + r, i = r + 1, j = r + 2;
        if (y >= pl.H) break;
        const uint64_t d = D[i];
        // erosion: out-of-image neighbours count as set
        const uint64_t dl = k > 0 ? ((L[j] >> 63) | (L[j] >> 62) | C[j] | (L[j - 1] >> 63) | (L[j + 1] >> 63)) & 1ull : 1ull;
        uint64_t dr = 0;                                           // right neighbour of the last column is outside
        if (valid > 64) dr = ((R[j] | (R[j] >> 1) | (C[j] >> 63) | R[j - 1] | R[j + 1]) & 1ull) << 63;
        uint64_t right = (d >> 1) | dr;
        if (valid <= 64) right |= 1ull << (valid - 1);             // pixel W-1 looks right at "outside" = set
        const uint64_t leftn = (d << 1) | dl;
        const uint64_t upn = y > 0 ? D[i - 1] : ~0ull;
        const uint64_t dnn = y + 1 < pl.H ? D[i + 1] : ~0ull;
        op[(size_t)y * pl.wpr + k] = d & leftn & right & upn & dnn & vmask;
    }
}

// flatten + per-root area (aux0) and brightness sum (aux1)
__global__ void cluster_accum_kernel(const uint64_t *bits, const uint8_t *frames, Plane pl, int count, int *par,
                                     int *aux0, unsigned *aux1)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    if (!rem) return;
    const size_t plane = g / pl.words;
    const int w = (int)(g - plane * pl.words);
    const int y = w / pl.wpr, k = w - y * pl.wpr;
    int *pp = par + plane * pl.ids;
    const uint8_t *row = frames + (plane * pl.H + y) * (size_t)pl.W + 64 * k;
    while (rem) {
        int s, len; uint64_t m;
        first_segment(rem, s, len, m);
        rem &= ~m;
        const int id = (w << 6) | s;
        const int root = find_root(pp, id);
        pp[id] = root;
        unsigned bs = 0;
        for (int i = 0; i < len; ++i) bs += row[s + i];
        atomicAdd(aux0 + plane * pl.ids + root, len);
        atomicAdd(aux1 + plane * pl.ids + root, bs);
    }
}

__device__ __forceinline__ int pix_at(const uint64_t *bp, const Plane &pl, int x, int y)
{
    if ((unsigned)x >= (unsigned)pl.W || (unsigned)y >= (unsigned)pl.H) return 0;
    return (int)((bp[(size_t)y * pl.wpr + (x >> 6)] >> (x & 63)) & 1);
}

// Outer border following from the component's raster-first pixel (findContours RETR_EXTERNAL,
// CHAIN_APPROX_NONE): returns the number of axis-aligned and diagonal moves (ip.py:691-693).
__device__ void trace_outer(const uint64_t *bp, const Plane &pl, int x0, int y0, int &n_axis, int &n_diag)
{
    const int dx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int dy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    n_axis = 0; n_diag = 0;
    int s = 4, i1x = x0, i1y = y0;
    do {
        s = (s - 1) & 7;
        i1x = x0 + dx[s]; i1y = y0 + dy[s];
    } while (!pix_at(bp, pl, i1x, i1y) && s != 4);
    if (s == 4) return;                               // single pixel: perimeter 0
    int i3x = x0, i3y = y0, i4x = x0, i4y = y0;
    for (int guard = 0; guard < (1 << 20); ++guard) {
        while (s < 15) {
            ++s;
            i4x = i3x + dx[s & 7]; i4y = i3y + dy[s & 7];
            if (pix_at(bp, pl, i4x, i4y)) break;
        }
        s &= 7;
        if (s & 1) ++n_diag; else ++n_axis;
        if (i4x == x0 && i4y == y0 && i3x == i1x && i3y == i1y) break;
        i3x = i4x; i3y = i4y;
        s = (s + 4) & 7;
    }
}

struct ClusterArgs {
    int min_size, max_size, circ_size;
    float min_circ;
    const double *pow2; int tdim;
};

// one lane per word; acts on the segments that are component roots
__global__ void cluster_eval_kernel(const uint64_t *bits, Plane pl, int count, const int *par, int *aux0,
                                    const unsigned *aux1, ClusterArgs ca, unsigned long long *stats)
{
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    if (!rem) return;
    const size_t plane = g / pl.words;
    const int w = (int)(g - plane * pl.words);
    const int y = w / pl.wpr, k = w - y * pl.wpr;
    const uint64_t *bp = bits + plane * pl.words;
    const int *pp = par + plane * pl.ids;
    while (rem) {
        int s, len; uint64_t m;
        first_segment(rem, s, len, m);
        rem &= ~m;
        const int id = (w << 6) | s;
        if (pp[id] != id) continue;
        const int area = aux0[plane * pl.ids + id];
        if (area < ca.min_size || area > ca.max_size) continue;      // ip.py:678
        bool kept = true;
        if (area >= ca.circ_size) {                                   // ip.py:685-695
            int na, nd;
            trace_outer(bp, pl, 64 * k + s, y, na, nd);
            float circ = 0.f;
            if (na + nd > 0) {
                const double per = (double)na + (double)nd * 1.4142135381698608;   // (double)sqrtf(2)
                const double p2 = (na < ca.tdim && nd < ca.tdim) ? ca.pow2[na * ca.tdim + nd] : per * per;
                const double q = (double)area / p2;
                circ = (float)(12.566370614359172 * q);                // 4*np.pi, stored to float32
            }
            kept = circ >= ca.min_circ;                                // ip.py:701 (float32 compare)
        }
        if (kept) {
            atomicAdd(stats + plane * 3 + 0, 1ull);
            atomicAdd(stats + plane * 3 + 1, (unsigned long long)area);
            atomicAdd(stats + plane * 3 + 2, (unsigned long long)aux1[plane * pl.ids + id]);
            aux0[plane * pl.ids + id] = -area;                         // mark for the cleaned mask
        }
    }
}

__global__ void select_marked_kernel(const uint64_t *bits, Plane pl, int count, const int *par, const int *aux0,
                                     uint64_t *out)
{
    IPX_VGPR_PAD_SAFE();
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= pl.words * count) return;
    uint64_t rem = bits[g];
    uint64_t r = 0;
    if (rem) {
        const size_t plane = g / pl.words;
        const int w = (int)(g - plane * pl.words);
        const int *pp = par + plane * pl.ids;
        const int *a0 = aux0 + plane * pl.ids;
        while (rem) {
            int s, len; uint64_t m;
            first_segment(rem, s, len, m);
            rem &= ~m;
            if (a0[pp[(w << 6) | s]] < 0) r |= m;
        }
    }
    out[g] = r;
}

Plane make_plane(int H, int W)
{
    Plane pl;
    pl.H = H; pl.W = W; pl.wpr = ipx_wpr(W);
    pl.words = (size_t)H * pl.wpr;
    pl.ids = pl.words * 64;
    return pl;
}

// planes per CCL batch so that the scratch stays near `budget` bytes
int ccl_batch(const Plane &pl, int count, size_t per_plane_bytes, size_t budget)
{
    size_t b = budget / per_plane_bytes;
    if (b < 1) b = 1;
    if (b > (size_t)count) b = count;
    return (int)b;
}

constexpr size_t CCL_BUDGET = (size_t)8 << 30;     // ~490 planes of 1080p per labelling batch

}  // namespace

// remove_bright_regions on `count` planes with the labelling scratch given by the caller of this function:
// `scr` holds B planes' worth of (parent, aux0, remove plane).  Nothing in it has to start at zero: ccl_init
// writes the parent / aux0 entry of every segment that exists in the plane at hand, later kernels touch only
// those entries, and select_big writes every word of the `remove` plane.  `profiled` = 0 launches without the
// built-in profiler (whose event pool grows by allocation).
#define CCL_LAUNCH(profiled, kid, bytes, kernel, grid, block, ...)                              \
    do {                                                                                       \
        if (profiled) IPX_LAUNCH(ctx, kid, bytes, kernel, grid, block, __VA_ARGS__);           \
        else hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, __VA_ARGS__);             \
    } while (0)

static int remove_bright_regions_run(ipx_ctx *ctx, const Plane &pl, void *scr, int B, int profiled,
                                     const uint64_t *bright_bits, const uint64_t *damaged_bits, int count,
                                     uint64_t *out_bits)
{
    int *par = (int *)scr;
    int *aux0 = par + pl.ids * B;
    uint64_t *remove = (uint64_t *)(aux0 + pl.ids * B);
    const int min_area = ctx->p.max_cluster_size;
    for (int first = 0; first < count; first += B) {
        const int nb = count - first < B ? count - first : B;
        const uint64_t *bb = bright_bits + first * pl.words;
        const uint64_t *db = damaged_bits + first * pl.words;
        uint64_t *ob = out_bits + first * pl.words;
        const size_t nw = pl.words * nb;
        const dim3 grid((unsigned)((nw + 255) / 256)), block(256);
        CCL_LAUNCH(profiled, KID_CCL_INIT, nw * 8, ccl_init_kernel, grid, block, bb, pl, nb, par, aux0, (unsigned *)nullptr);
        CCL_LAUNCH(profiled, KID_CCL_MERGE, nw * 8, ccl_merge_kernel, grid, block, bb, pl, nb, par);
        CCL_LAUNCH(profiled, KID_CCL_REDUCE, nw * 8, ccl_area_kernel, grid, block, bb, pl, nb, par, aux0);
        CCL_LAUNCH(profiled, KID_CCL_SELECT, nw * 16, select_big_kernel, grid, block, bb, pl, nb, par, aux0, min_area, remove);
        CCL_LAUNCH(profiled, KID_CCL_INIT, nw * 8, ccl_init_kernel, grid, block, db, pl, nb, par, aux0, (unsigned *)nullptr);
        CCL_LAUNCH(profiled, KID_CCL_MERGE, nw * 8, ccl_merge_kernel, grid, block, db, pl, nb, par);
        CCL_LAUNCH(profiled, KID_CCL_REDUCE, nw * 16, ccl_taint_kernel, grid, block, db, remove, pl, nb, par, aux0);
        CCL_LAUNCH(profiled, KID_CCL_SELECT, nw * 16, select_untainted_kernel, grid, block, db, pl, nb, par, aux0, ob);
        IPX_LAUNCH_CHECK();
    }
    return IPX_OK;
}

// per plane: parent + aux0 (int32 per id) + one `remove` plane
static inline size_t rbr_bytes_per_plane(const Plane &pl) { return pl.ids * 8 + pl.words * 8; }

extern "C" int ipx_remove_bright_regions(ipx_ctx *ctx, const uint64_t *bright_bits, const uint64_t *damaged_bits,
                                         int count, int H, int W, uint64_t *out_bits)
{
    IPX_REQUIRE(ctx && bright_bits && damaged_bits && out_bits, "null pointer");
    IPX_REQUIRE(count > 0 && H > 0 && W > 0, "bad sizes");
    IPX_REQUIRE((size_t)H * ipx_wpr(W) * 64 < ((size_t)1 << 31), "frame too large for 32-bit segment ids");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const Plane pl = make_plane(H, W);
    const size_t per_plane = rbr_bytes_per_plane(pl);
    const int B = ccl_batch(pl, count, per_plane, CCL_BUDGET);
    void *scr;
    if (int e = ipx_scratch(ctx, SLOT_CCL, per_plane * B, &scr)) return e;
    return remove_bright_regions_run(ctx, pl, scr, B, 1, bright_bits, damaged_bits, count, out_bits);
}

// ---- the entry point that allocates nothing ----------------------------------------------------------------
// Labelling scratch of ipx_remove_bright_regions_noalloc: one array per GPU the library runs on, held until the
// process ends.  In host code the NAME below is only a handle for hipGetSymbolAddress: the device address is
// resolved once per device by ipx_static_scratch_init (called from ipx_ctx_create with that device current) and
// from then on only that pointer is used.
static __device__ __align__(256) unsigned char scratch[IPX_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
void *g_scratch_ptr[MAX_DEVICES];          // device address of `scratch` on device d, or null
}

// Called by ipx_ctx_create with `device` current.  Loading the code object (and with it the array) on a device
// happens here at the latest, not inside the entry point.
int ipx_static_scratch_init(int device)
{
    if (device < 0 || device >= MAX_DEVICES) return IPX_OK;           // such a device gets no no-alloc entry point
    void *ptr = __atomic_load_n(&g_scratch_ptr[device], __ATOMIC_ACQUIRE);
    if (ptr) return IPX_OK;
    IPX_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    __atomic_store_n(&g_scratch_ptr[device], ptr, __ATOMIC_RELEASE);   // every thread resolves the same address
    return IPX_OK;
}

extern "C" size_t ipx_noalloc_scratch_bytes(void) { return IPX_NOALLOC_SCRATCH_BYTES; }

extern "C" int ipx_remove_bright_regions_noalloc(ipx_ctx *ctx, const uint64_t *bright_bits, const uint64_t *damaged_bits,
                                                 int count, int H, int W, uint64_t *out_bits)
{
    IPX_REQUIRE(ctx && bright_bits && damaged_bits && out_bits, "null pointer");
    IPX_REQUIRE(count > 0 && H > 0 && W > 0, "bad sizes");
    const Plane pl = make_plane(H, W);
    const size_t per_plane = rbr_bytes_per_plane(pl);
    if (per_plane > IPX_NOALLOC_SCRATCH_BYTES) {
        ipx_set_error("ipx_remove_bright_regions_noalloc: a %d x %d plane needs %zu bytes of labelling scratch, the static "
                      "array holds %zu (planes of up to %d 64-pixel words, e.g. 3840 x 2160); use ipx_remove_bright_regions",
                      W, H, per_plane, (size_t)IPX_NOALLOC_SCRATCH_BYTES, IPX_NOALLOC_MAX_WORDS);
        return IPX_ERR_UNSUPPORTED;
    }
    void *scr = (ctx->device >= 0 && ctx->device < MAX_DEVICES) ? __atomic_load_n(&g_scratch_ptr[ctx->device], __ATOMIC_ACQUIRE) : nullptr;
    if (!scr) {
        ipx_set_error("ipx_remove_bright_regions_noalloc: no static scratch on device %d", ctx->device);
        return IPX_ERR_UNSUPPORTED;
    }
    if (int e = ipx_ctx_enter(ctx)) return e;
    size_t B = IPX_NOALLOC_SCRATCH_BYTES / per_plane;
    if (B > (size_t)count) B = count;
    return remove_bright_regions_run(ctx, pl, scr, (int)B, 0, bright_bits, damaged_bits, count, out_bits);
}

extern "C" int ipx_cluster_stats(ipx_ctx *ctx, const uint8_t *frames, const uint64_t *mask_bits, const uint8_t *keep,
                                 int n, int H, int W, const double *pow2_table, int tdim,
                                 int64_t *stats, uint64_t *cleaned_bits)
{
    IPX_REQUIRE(ctx && frames && mask_bits && stats, "null pointer");
    IPX_REQUIRE(n > 0 && H > 0 && W > 0, "bad sizes");
    IPX_REQUIRE(pow2_table == nullptr || tdim > 0, "bad pow2 table");
    IPX_REQUIRE((size_t)H * ipx_wpr(W) * 64 < ((size_t)1 << 31), "frame too large for 32-bit segment ids");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const Plane pl = make_plane(H, W);
    // per plane: parent + aux0 + aux1 (4 bytes per id each) + the closed plane
    const size_t per_plane = pl.ids * 12 + pl.words * 8;
    int B = ccl_batch(pl, n, per_plane, CCL_BUDGET);
    if (B > 65535) B = 65535;                            // close_cross3_kernel takes the batch's planes in grid.z
    void *scr;
    if (int e = ipx_scratch(ctx, SLOT_CCL, per_plane * B, &scr)) return e;
    int *par = (int *)scr;
    int *aux0 = par + pl.ids * B;
    unsigned *aux1 = (unsigned *)(aux0 + pl.ids * B);
    uint64_t *closed = (uint64_t *)(aux1 + pl.ids * B);
    ClusterArgs ca;
    ca.min_size = ctx->p.min_cluster_size;
    ca.max_size = ctx->p.max_cluster_size;
    ca.circ_size = ctx->p.circularity_size_threshold;
    ca.min_circ = (float)ctx->p.min_circularity;
    ca.pow2 = pow2_table;
    ca.tdim = pow2_table ? tdim : 0;
    IPX_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(int64_t) * 3 * (size_t)n, ctx->stream));
    for (int first = 0; first < n; first += B) {
        const int nb = n - first < B ? n - first : B;
        const uint64_t *mb = mask_bits + first * pl.words;
        const uint8_t *fr = frames + (size_t)first * H * W;
        const uint8_t *kp = keep ? keep + first : nullptr;
        const size_t nw = pl.words * nb;
        const dim3 grid((unsigned)((nw + 255) / 256)), block(256);
        IPX_LAUNCH(ctx, KID_CLOSE, nw * 16, close_cross3_kernel, dim3((pl.wpr + 31) / 32, (pl.H + 8 * CC_RB - 1) / (8 * CC_RB), nb), dim3(256), mb, pl, nb, kp, closed);
        IPX_LAUNCH(ctx, KID_CCL_INIT, nw * 8, ccl_init_kernel, grid, block, closed, pl, nb, par, aux0, aux1);
        IPX_LAUNCH(ctx, KID_CCL_MERGE, nw * 8, ccl_merge_kernel, grid, block, closed, pl, nb, par);
        IPX_LAUNCH(ctx, KID_CCL_REDUCE, nw * 8, cluster_accum_kernel, grid, block, closed, fr, pl, nb, par, aux0, aux1);
        IPX_LAUNCH(ctx, KID_CLUSTER_EVAL, nw * 8, cluster_eval_kernel, grid, block, closed, pl, nb, par, aux0, aux1, ca,
                           (unsigned long long *)stats + (size_t)first * 3);
        if (cleaned_bits)
            IPX_LAUNCH(ctx, KID_CCL_SELECT, nw * 16, select_marked_kernel, grid, block, closed, pl, nb, par, aux0,
                               cleaned_bits + first * pl.words);
        IPX_LAUNCH_CHECK();
    }
    return IPX_OK;
}

// detect.hip -- temporal background + brightness/contrast tests -> packed bit-planes (gfx950).
//
// Replaces, for a resident chunk of uint8 frames:
//   compute_background/_find_background  ip.py:183-252  (numba typing: SURVEY App. A.1)
//   get_damaged_pixel_mask               ip.py:635-648  (f32 threshold chain A.2, 30x30 box mean A.3)
//   (background > brightness_threshold)  ip.py:743      (input of remove_bright_regions)
//
// One 256-thread workgroup owns a 128 x 32 pixel tile of one frame (128 pixels = one 128-byte line per row of
// every temporal neighbour).  The 30x30 window sum is an exact integer: the (128+29) x (32+29) window is staged
// in LDS, summed horizontally (u16) into LDS, and every thread slides the vertical sum down its own 4 columns
// x 4 rows in registers.  `frame > S*(1/900.0)` in double is equivalent to `900*frame > S` in integers
// (S*(1/900.0) == v exactly when S == 900 v, and the next integers are 1/900 away).  Mask bits leave the kernel
// as 64-bit words assembled from per-thread nibbles.
//
// Workgroups are numbered so that an XCD (its own 4 MiB L2; ids are dealt round-robin) owns a fixed eighth of
// the frame's tiles for ALL frames and walks them frame by frame: the six neighbour frames a tile reads were
// fetched into that L2 when they were centre / neighbour of the frames just before, so every frame byte
// crosses the fabric about once instead of seven times.
//
// Floating-point contract (must stay bit-exact with the reference): no FMA contraction, IEEE f32 divide, IEEE
// f64 divide/sqrt -- the library is built with -ffp-contract=off and without fast-math.  The outlier cut of
// _find_background is DECIDED in integers wherever that is provably the reference's decision (background_px),
// and by the reference's float chain (background_px_chain) in the rare cases where it is not.
#include "ipx_internal.h"

namespace {

// Tile height 32 (round 3; 64 before): 29.5 KB of LDS and 85 VGPRs per workgroup put five workgroups on a CU instead of three,
// which the scan (short, latency-bound) needs more than it minds the taller halo: 3.91 against 4.21 ms per 1000 frames of 1080p,
// an all-bright scene 6.97 against 7.90 ms (A/B on one box, -DIPX_DETECT_TH=64 builds the other).
#ifndef IPX_DETECT_TH
#define IPX_DETECT_TH 32
#endif
constexpr int TW = 128, TH = IPX_DETECT_TH, KS = 30, AX = 15;
static_assert(TH == 64 || TH == 32, "tile height: 256 threads own 4 columns x TH/8 rows each");
constexpr int WIN_H = TH + KS - 1;       // 61 rows
constexpr int WIN_W = TW + 32;           // staged columns x0-16 .. x0+143 (16-byte aligned on both sides)
constexpr int WIN_STRIDE = WIN_W + 16;   // bytes per staged row (176: 16-byte aligned, 44 dwords: rows 12 banks apart)
constexpr int HS_STRIDE = TW;            // u16 per row of horizontal sums
constexpr int RG = TH / 8;               // rows per thread in the per-pixel phase (4 columns x 8 rows at TH = 64)

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

// _find_background for one pixel, the reference's own float chain: n temporal neighbours
// b[0..n), rcp[i] = 1.0 / i.  Only reached for the pixels background_px cannot decide in integers.
template <int MAXN>
__device__ __forceinline__ float background_px_chain(const int (&b)[MAXN], int n, double coeff, const double *rcp)
{
    const double yn = rcp[n];
    int S = 0;
#pragma unroll
    for (int k = 0; k < MAXN; ++k)
        if (k < n) S += b[k];
    const float mean = (float)((double)S * yn);  // == (float)S / (float)n == f32(f64(S)/n)
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < MAXN; ++k)
        if (k < n) {
            const float diff = (float)b[k] - mean;
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
        if (k < n && b[k] <= T) { Si += b[k]; ++c; }
    return c > 0 ? (float)((double)Si * rcp[c]) : mean;
}

// Per-workgroup constants of the integer decision below.
struct BgConst {
    int n;            // neighbours of this frame (centre excluded)
    int nodd;         // odd part of n: S/n is a dyadic rational (exact in f32) iff nodd divides S
    unsigned magic;   // ceil(2^16 / nodd): floor(S / nodd) == (S * magic) >> 16 for S < 4096, nodd <= 15
    bool unit_coeff;  // pixel_std_coeff == 1.0 (the reference's default, ip.py:204)
};

// Correctly rounded f32 quotient a / b for an integer-valued a and a small integer b, y = RN32(1 / b): one
// residual correction of q = RN32(a y) (the residual a - b q is exact in an fma).  Exact over the whole domain
// this kernel uses it on -- every Si / c with c <= 16, Si <= 255 c, and every such quotient divided by 255 --
// which tests/test_abi_and_host.py walks through.
__device__ __forceinline__ float div_small(float a, float b, float y)
{
    const float q = a * y;
    const float r = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(r, y, q);
}

// _find_background for one pixel.  The reference keeps neighbour b iff  b <= mean32 + sd32  (coeff 1; the f64
// sum of the two f32 values is exact).  With S = sum b, Q = sum b^2, D = n Q - S^2 (= n^2 variance) and
// t = n b - S (= n (b - mean)) that is, in exact arithmetic,  t <= 0  or  t^2 <= D:  integers below 2^24.  The
// reference's roundings move the two sides only slightly: mean32 is off by <= 2^-17, so n (b - mean32) = t (1 + r)
// with |r| <= 2^-17 n / t; and sd32^2 = (D / n^2 + e^2)(1 + h) with e <= 2^-17, |h| <= 5 * 2^-24 (+2^-50) from the
// three f32 roundings of the squares' path, the f32 rounding of sd and the f64 division and root.  Hence
//    reference keeps b   if  t^2 + 2^-16 n t + 2^-21.5 D <= D,      drops b  if  t^2 - 2^-16 n t > D + 2^-21.5 D + 2^-26
// and with t <= 255 n, D <= n^2 255^2 / 4 the slack terms sum to < 0.35 for n <= 6 and < 2.45 for n <= 16:
// |t^2 - D| >= M (M = 1 resp. 3) decides exactly as the reference does; t <= 0 always keeps (sd32 > 2^-17 when
// D >= 1, and D = 0 means all values equal).  A tie t^2 == D is the reference's too when S / n is dyadic: then
// mean32, every difference, every square (<= 24 bits), their sum, sum / n = (t / n)^2 and its root t / n are all
// exact and mean32 + sd32 == b: kept.  What is left (a tie with an inexact mean, |t^2 - D| < M for n > 6, any
// coeff other than 1) goes through the reference's float chain.  tools/fuzz_stages.py and the parity sweeps of
// tests/test_gpu_stages.py compare the result with the oracle's chain.
//
// Written as arithmetic on 24-bit integers (no boolean network: one multiply-add, one max, one multiply-add, one
// shift, one multiply-add, one add and one min per neighbour).
// 24-bit multiply-adds, forced: the compiler otherwise proves the ranges only sometimes and falls back to the
// quarter-rate 32/64-bit multiplier.  Every operand below is within +-2^23 by construction (bytes, n <= 16,
// S <= 4080, Q <= 16 * 255^2, t <= 4080).
__device__ __forceinline__ int mad_i24(int a, int b, int c)
{
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct BgSums { unsigned Si, c; bool hard; };

// s + byte K of d / s - byte K of d in one instruction (sub-dword operand select)
template <int K>
__device__ __forceinline__ int add_byte(int s, uint32_t d)
{
    int r;
    if (K == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(s), "v"(d));
    else if (K == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(s), "v"(d));
    else if (K == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(s), "v"(d));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(s), "v"(d));
    return r;
}
template <int K>
__device__ __forceinline__ int sub_byte(int s, uint32_t d)
{
    int r;
    if (K == 0) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(s), "v"(d));
    else if (K == 1) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(s), "v"(d));
    else if (K == 2) asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(s), "v"(d));
    else asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(s), "v"(d));
    return r;
}

// one step of the horizontal slide: s(t) = s(t-1) + byte(30 + t) - byte(t), bytes numbered through d[0..11]
template <int T>
__device__ __forceinline__ int hslide(int s, const uint32_t (&d)[12])
{
    return sub_byte<T & 3>(add_byte<(30 + T) & 3>(s, d[(30 + T) >> 2]), d[T >> 2]);
}

// number of the four bytes of x that are > T, for a constant T prepared by swar_gt_prepare
struct SwarGt { uint32_t add; int mode; };     // mode 0: none, 1: all, 2: T < 128, 3: T >= 128
__device__ __forceinline__ SwarGt swar_gt_prepare(int T)
{
    SwarGt g;
    g.mode = T >= 255 ? 0 : (T < 0 ? 1 : (T < 128 ? 2 : 3));
    g.add = (uint32_t)(0x7f - (T & 0x7f)) * 0x01010101u;
    return g;
}
__device__ __forceinline__ unsigned swar_count_gt(uint32_t x, const SwarGt &g)
{
    if (g.mode == 0) return 0;
    if (g.mode == 1) return 4;
    const uint32_t low = (x & 0x7f7f7f7fu) + g.add;         // bit 7 of a byte: (x & 127) > (T & 127); no carry between bytes
    const uint32_t m = (g.mode == 2 ? (low | x) : (low & x)) & 0x80808080u;
    return __builtin_popcount(m);
}

template <int MAXN, bool FULL>
__device__ __forceinline__ BgSums background_sums(const uint32_t (&v)[MAXN], int j, const BgConst &bc)
{
    const int n = FULL ? MAXN : bc.n;
    int b[MAXN];
    int S = 0, Q = 0;
#pragma unroll
    for (int k = 0; k < MAXN; ++k) {
        b[k] = (int)__builtin_amdgcn_ubfe(v[k], 8 * j, 8);     // slots k >= n hold 0
        S += b[k];
        Q = k == 0 ? (int)__umul24(b[k], b[k]) : (int)mad_u24(b[k], b[k], Q);
    }
    const int negS = -S;
    const int negD1 = mad_i24(S, S, mad_i24(Q, -n, -1));        // -(D + 1), D = n Q - S^2
    BgSums r;
    r.Si = 0; r.c = 0; r.hard = false;
    unsigned z = 0;                                              // all ones when some kept neighbour ties (u^2 == D)
#pragma unroll
    for (int k = 0; k < MAXN; ++k) {
        if (FULL || k < n) {                                    // workgroup-uniform
            const int t = mad_i24(b[k], n, negS);
            const int u = max(t, 0);
            const int w = mad_i24(u, u, negD1);                 // w < 0  <=>  u^2 <= D;  w == -1  <=>  u^2 == D
            const unsigned m = (unsigned)w >> 31;
            r.Si = mad_u24(b[k], m, r.Si);
            r.c += m;
            z = max(z, (unsigned)w);
            if (MAXN > 6) r.hard |= t > 0 && (unsigned)(w + 3) <= 4u && w != -1;   // 0 < |u^2 - D| < 3
        }
    }
    // a tie is exact only when S / n is dyadic (branch-free: the test is three instructions)
    const int q = (int)(((unsigned)S * bc.magic) >> 16);
    r.hard |= z == ~0u && S - q * bc.nodd != 0;
    return r;
}

// the reference's float chain for pixel j of the four in v (the pixels background_sums could not decide)
template <int MAXN> struct NeighbourRow { uint32_t v[MAXN]; };      // by value: travels in registers, not scratch
template <int MAXN>
__device__ __noinline__ float background_hard(NeighbourRow<MAXN> row, int j, int n, double coeff, const double *rcp)
{
    int b[MAXN];
#pragma unroll
    for (int k = 0; k < MAXN; ++k) b[k] = (int)((row.v[k] >> (8 * j)) & 0xff);
    return background_px_chain<MAXN>(b, n, coeff, rcp);
}

// ip.py:638: 30 + (bg / 255) * 225, three separately rounded f32 operations, then maximum(30, .).  bg is a
// quotient Si / c (see div_small for why the division below is the IEEE one on that domain).
__device__ __forceinline__ float damage_threshold_f32(float bg)
{
    float t = div_small(bg, 255.0f, 1.0f / 255.0f);
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
    int n, H, W, first, radius, count;
    int tiles_x, tiles_y, per_xcd;   // tiles of a frame and how many of them one XCD owns
    int frame_gt, est_lt;
    float bg_gt;
    double coeff;
};

// 16 nibbles held one per byte (four dwords) -> 64 bits, nibble k at bits 4k..4k+3
__device__ __forceinline__ uint32_t nibbles4(uint32_t d)
{
    d = (d | (d >> 4)) & 0x00ff00ffu;
    return (d | (d >> 8)) & 0xffffu;
}

// ---- the row pre-test (FAST tiles) ----------------------------------------------------------------------------------
// Most pixels of most footage can be neither "damaged" nor "bright background", and that can be PROVED from the smallest
// and the largest of a pixel's temporal neighbours, without the outlier cut:
//   * the background is a mean of a non-empty subset of the neighbours (or of all of them), rounded to f32: it lies in
//     [min b, max b], whichever subset the reference's cut keeps and whatever pixel_std_coeff is;
//   * `background > brightness_threshold` (ip.py:743) therefore needs  max b > brightness_threshold;
//   * the damage threshold 30 + (bg / 255) * 225 (ip.py:638; the constants are the reference's own) does not decrease with
//     bg, so `pixel > threshold` needs  pixel > 30 + (min b / 255) * 225, and the three f32 roundings move the right side by
//     less than 1e-4: in integers, 255 pixel >= 7649 + 225 min b is NECESSARY (one unit of 1/255 of slack).
// Both are evaluated for four pixels at once on 16-bit lanes (every quantity is below 2^16): even / odd bytes of each
// neighbour dword, packed min / max, one packed multiply-add and one saturating subtraction per test.  Only if some lane
// of the wave has a pixel that passes one of the two does the wave run the outlier cut for that row (bit-exact as before:
// the pre-test only ever skips pixels whose two result bits are provably zero).  On the benchmark's footage 2 % of the
// wave-rows run the cut; an all-bright scene runs it everywhere and pays the pre-test on top (+12 %).
typedef unsigned short ipx_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ipx_u16x2 even_bytes(uint32_t v) { return __builtin_bit_cast(ipx_u16x2, v & 0x00ff00ffu); }
__device__ __forceinline__ ipx_u16x2 odd_bytes(uint32_t v) { return __builtin_bit_cast(ipx_u16x2, __builtin_amdgcn_perm(v, v, 0x0c030c01u)); }

struct PreTest {
    ipx_u16x2 k255, k225, k7648, kbright;      // constants of the two tests, in both halves
    bool on;                                   // brightness_threshold in [0, 65535): else every row runs the cut
};
__device__ __forceinline__ PreTest pretest_prepare(float bg_gt)
{
    PreTest t;
    t.on = bg_gt >= 0.f && bg_gt < 65535.f;
    const unsigned short kb = t.on ? (unsigned short)floorf(bg_gt) : (unsigned short)0;   // max b > bg_gt  <=>  max b > floor(bg_gt)
    t.k255 = ipx_u16x2{255, 255}; t.k225 = ipx_u16x2{225, 225}; t.k7648 = ipx_u16x2{7648, 7648}; t.kbright = ipx_u16x2{kb, kb};
    return t;
}
// nonzero iff one of the four pixels of `cen` may get a result bit, given its neighbours' packed min / max
__device__ __forceinline__ uint32_t pretest_four(const PreTest &t, uint32_t cen, ipx_u16x2 mnE, ipx_u16x2 mnO, ipx_u16x2 mxE, ipx_u16x2 mxO)
{
    const ipx_u16x2 dE = __builtin_elementwise_sub_sat(even_bytes(cen) * t.k255, mnE * t.k225 + t.k7648);   // > 0: 255 p >= 7649 + 225 min
    const ipx_u16x2 dO = __builtin_elementwise_sub_sat(odd_bytes(cen) * t.k255, mnO * t.k225 + t.k7648);
    const ipx_u16x2 bE = __builtin_elementwise_sub_sat(mxE, t.kbright), bO = __builtin_elementwise_sub_sat(mxO, t.kbright);
    return (__builtin_bit_cast(uint32_t, dE) | __builtin_bit_cast(uint32_t, dO)) | (__builtin_bit_cast(uint32_t, bE) | __builtin_bit_cast(uint32_t, bO));
}

// Phase C of detect_kernel for one thread (see there).  FULL: the frame has all MAXN temporal neighbours.
// FAST: the neighbour rows are read as aligned dwords at a 32-bit offset from a per-frame base that is uniform
// over the workgroup (scalar registers), one add per row; no per-pixel bounds tests.
//
// FAST tiles first SCAN: the row pre-test (pretest_four) on every group of four pixels, no outlier cut.  A group that
// passes the pre-test (on the benchmark's footage 1.4 % of them: a hot pixel, or one of the six frames around a hot
// pixel, or the bright block) is appended to a work list in LDS, and when the scan is done the workgroup's threads take
// the list's entries one each -- dense lanes for the expensive code instead of a wave that runs it for 64 lanes because
// one of them needs it.  A tile whose list overflows (bright background) runs the cut for all of its groups in the
// per-row loop instead, as tiles without the pre-test do: the dense case, which then costs the scan on top (+15 %).
constexpr int WORK_CAP = 512;                // groups of four pixels per tile that can wait in the list (of 1024)
// rows of neighbour dwords the scan keeps in flight (the 16-neighbour variant has registers for one)
template <int MAXN> struct SCAN_AHEAD { static constexpr int value = MAXN > 6 ? 1 : (RG < 4 ? RG : 4); };

template <int MAXN, bool FULL, bool FAST>
__device__ __forceinline__ void pixel_rows(const DetectArgs &a, const uint8_t *s_win, const uint16_t *s_hs,
                                           uint8_t (*s_nib)[TH][TW / 4], const double *s_rcp, const float *s_rcpf,
                                           unsigned int *s_nwork, unsigned short *s_work,
                                           int x0, int y0, int fi, int lo, int nn, int cpos,
                                           unsigned int &c_lt, unsigned int &c_gt)
{
    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const bool aligned = (W & 3) == 0;
    BgConst bc;
    bc.n = nn;
    bc.nodd = nn > 0 ? nn >> __builtin_ctz(nn) : 1;
    bc.magic = (65536u + bc.nodd - 1) / bc.nodd;
    bc.unit_coeff = a.coeff == 1.0;
    const int cg = tid & 31, rg = tid >> 5;
    const int gx = x0 + 4 * cg;
    const bool col_live = gx < W;
    // the 30-row sums of the window whose top row is `top`, for the four columns of group `grp`
    auto vsum_at = [&](int top, int grp, int (&vs)[4]) {
        vs[0] = vs[1] = vs[2] = vs[3] = 0;
        // 30 rows of packed u16 pairs: up to 8 rows (8 * 30 * 255 < 2^16) are added as whole dwords before widening
        const uint16_t *hp = s_hs + top * HS_STRIDE + 4 * grp;
#pragma unroll
        for (int g = 0; g < KS; g += 8) {
            uint32_t px = 0, py = 0;
#pragma unroll
            for (int i = g; i < (g + 8 < KS ? g + 8 : KS); ++i) {
                const uint2 h = *reinterpret_cast<const uint2 *>(hp + i * HS_STRIDE);
                px += h.x; py += h.y;
            }
            vs[0] += px & 0xffff; vs[1] += px >> 16; vs[2] += py & 0xffff; vs[3] += py >> 16;
        }
    };
    int vs[4];
    const SwarGt sw_gt = swar_gt_prepare(a.frame_gt), sw_ge = swar_gt_prepare(a.est_lt - 1);
    const PreTest pt = pretest_prepare(a.bg_gt);
    const bool listed = FAST && pt.on;                       // workgroup-uniform: the pre-test and the work list are in use
    const size_t P = (size_t)H * W;
    // neighbour frames of one row of group column `x`: slot s holds frame lo + s (+1 from the centre on)
    auto load_row = [&](int gy, int x, uint32_t (&v)[MAXN]) {
        if (FAST) {
            const unsigned off = (unsigned)(min(gy, H - 1) * W + x);
#pragma unroll
            for (int s = 0; s < MAXN; ++s) {
                const uint8_t *fb = a.frames + (size_t)(lo + s + (s >= cpos ? 1 : 0)) * P;    // uniform
                v[s] = (FULL || s < nn) ? *reinterpret_cast<const uint32_t *>(fb + off) : 0u;
            }
        } else {
#pragma unroll
            for (int s = 0; s < MAXN; ++s) {
                const int f = lo + s + (s >= cpos ? 1 : 0);
                v[s] = (s < nn && col_live && gy < H) ? load4(a.frames + ((size_t)f * H + gy) * W, x, W, aligned) : 0u;
            }
        }
    };
    // the outlier cut and the two tests for the four pixels of `cen` (neighbour dwords v, window sums ws) -> two nibbles.
    // One pixel per trip of a rolled loop: the cut runs for listed groups and dense tiles only (see above), and rolled it
    // holds one pixel's neighbours, sums and decisions in registers instead of four -- which is what lets five waves per
    // SIMD run the scan.
    auto cut_four = [&](const uint32_t (&v)[MAXN], uint32_t cen, const int (&ws)[4], int gy, int x, uint32_t &dm, uint32_t &bm,
                        unsigned &lt, unsigned &gt) __attribute__((always_inline)) {   // (called by reference to register arrays)
        int w0 = ws[0], w1 = ws[1], w2 = ws[2], w3 = ws[3];     // handed down one place per trip: no indexed register array
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            if (!(FAST || x + j < W)) break;
            const BgSums sm = background_sums<MAXN, FULL>(v, j, bc);
            float bg = div_small((float)sm.Si, (float)sm.c, s_rcpf[sm.c]);   // c >= 1
            if (__builtin_expect(sm.hard || !bc.unit_coeff, 0)) {
                NeighbourRow<MAXN> row;
#pragma unroll
                for (int s = 0; s < MAXN; ++s) row.v[s] = v[s];
                bg = background_hard<MAXN>(row, j, FULL ? MAXN : nn, a.coeff, s_rcp);
            }
            const int pix = (int)__builtin_amdgcn_ubfe(cen, 8 * j, 8);
            if (!FAST && a.bg_out) a.bg_out[(size_t)gy * W + x + j] = bg;
            const float thr = damage_threshold_f32(bg);
            const bool dmg = ((float)pix > thr) && (900 * pix > w0);
            w0 = w1; w1 = w2; w2 = w3;
            dm |= (uint32_t)dmg << j;
            bm |= (uint32_t)(bg > a.bg_gt) << j;
            if (!FAST) { lt += pix < a.est_lt; gt += pix > a.frame_gt; }
        }
    };
    // the per-row loop that runs the cut for every group (tiles without the pre-test, and tiles whose list overflowed)
    auto dense_rows = [&]() {
        uint32_t vnext[MAXN];
        load_row(y0 + RG * rg, gx, vnext);
#pragma unroll 1
        for (int r = 0; r < RG; ++r) {
            const int ty = RG * rg + r;
            const int gy = y0 + ty;
            uint32_t v[MAXN];
#pragma unroll
            for (int s = 0; s < MAXN; ++s) v[s] = vnext[s];
            if (r + 1 < RG) load_row(gy + 1, gx, vnext);     // in flight under this row's arithmetic
            uint32_t dm = 0, bm = 0;
            if (FAST || (gy < H && col_live)) {
                const uint32_t cen = *reinterpret_cast<const uint32_t *>(s_win + (ty + AX) * WIN_STRIDE + 16 + 4 * cg);
                unsigned lt = 0, gt = 0;
                cut_four(v, cen, vs, gy, gx, dm, bm, lt, gt);
                if (!listed) {                               // (a listed tile has counted its pixels in the scan)
                    if (FAST) { gt = swar_count_gt(cen, sw_gt); lt = 4 - swar_count_gt(cen, sw_ge); }
                    if (!FAST || gy < H) { c_lt += lt; c_gt += gt; }
                }
            }
            s_nib[0][ty][cg] = (uint8_t)dm;
            s_nib[1][ty][cg] = (uint8_t)bm;
            if (r + 1 < RG) {                                // slide the window one row down
                const uint2 hin = *reinterpret_cast<const uint2 *>(s_hs + (ty + KS) * HS_STRIDE + 4 * cg);
                const uint2 hout = *reinterpret_cast<const uint2 *>(s_hs + ty * HS_STRIDE + 4 * cg);
                const ipx_u16x2 dx = __builtin_bit_cast(ipx_u16x2, hin.x) - __builtin_bit_cast(ipx_u16x2, hout.x);   // per lane, mod 2^16
                const ipx_u16x2 dy = __builtin_bit_cast(ipx_u16x2, hin.y) - __builtin_bit_cast(ipx_u16x2, hout.y);
                vs[0] += (int)(short)dx.x; vs[1] += (int)(short)dx.y;
                vs[2] += (int)(short)dy.x; vs[3] += (int)(short)dy.y;
            }
        }
    };
    if (!listed) {
        vsum_at(RG * rg, cg, vs);
        dense_rows();
        return;
    }
    // ---- the scan: pre-test of the thread's groups (one per row it owns), the frame's pixel counters, zero nibbles.  Nothing in it is long
    // enough to hide a load, so the rows' loads run four rows ahead (the loop is unrolled: the ring of rows is indexed
    // statically).  (Requesting the first four rows before the window is staged, so that their trip to memory runs under
    // phases A and B, made the kernel 6 % SLOWER, measured A/B on one box: 4.39 against 4.14 ms.)
    {
        constexpr int AHEAD = SCAN_AHEAD<MAXN>::value;
        uint32_t ring[AHEAD][MAXN];
#pragma unroll
        for (int r = 0; r < AHEAD; ++r) load_row(y0 + RG * rg + r, gx, ring[r]);
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int ty = RG * rg + r;
            const int gy = y0 + ty;
            uint32_t v[MAXN];
#pragma unroll
            for (int s = 0; s < MAXN; ++s) v[s] = ring[r % AHEAD][s];
            if (r + AHEAD < RG) load_row(gy + AHEAD, gx, ring[r % AHEAD]);
            const uint32_t cen = *reinterpret_cast<const uint32_t *>(s_win + (ty + AX) * WIN_STRIDE + 16 + 4 * cg);
            ipx_u16x2 mnE = even_bytes(v[0]), mnO = odd_bytes(v[0]), mxE = mnE, mxO = mnO;
#pragma unroll
            for (int s = 1; s < MAXN; ++s)
                if (FULL || s < nn) {                        // workgroup-uniform
                    const ipx_u16x2 e = even_bytes(v[s]), o = odd_bytes(v[s]);
                    mnE = __builtin_elementwise_min(mnE, e); mnO = __builtin_elementwise_min(mnO, o);
                    mxE = __builtin_elementwise_max(mxE, e); mxO = __builtin_elementwise_max(mxO, o);
                }
            if (pretest_four(pt, cen, mnE, mnO, mxE, mxO) != 0u) {               // rare, divergent
                const unsigned slot = atomicAdd(s_nwork, 1u);
                if (slot < (unsigned)WORK_CAP) s_work[slot] = (unsigned short)((ty << 5) | cg);
            }
            if (gy < H) { c_gt += swar_count_gt(cen, sw_gt); c_lt += 4 - swar_count_gt(cen, sw_ge); }
            s_nib[0][ty][cg] = 0;
            s_nib[1][ty][cg] = 0;
        }
    }
    __syncthreads();                                         // the list is complete, every nibble has its zero
    const unsigned nlisted = *s_nwork;
    if (nlisted > (unsigned)WORK_CAP) {                      // workgroup-uniform: a dense tile (bright background): every group, in place
        vsum_at(RG * rg, cg, vs);
        dense_rows();
        return;
    }
    // ---- the listed groups, one per thread: neighbour rows again (cache hits), window sums from the horizontal sums
    for (int i = tid; i < (int)nlisted; i += 256) {
        const int item = s_work[i], ty = item >> 5, grp = item & 31;
        const int gy = y0 + ty, x = x0 + 4 * grp;
        uint32_t v[MAXN];
        load_row(gy, x, v);
        int ws[4];
        vsum_at(ty, grp, ws);
        const uint32_t cen = *reinterpret_cast<const uint32_t *>(s_win + (ty + AX) * WIN_STRIDE + 16 + 4 * grp);
        uint32_t dm = 0, bm = 0;
        unsigned lt = 0, gt = 0;
        cut_four(v, cen, ws, gy, x, dm, bm, lt, gt);
        s_nib[0][ty][grp] = (uint8_t)dm;
        s_nib[1][ty][grp] = (uint8_t)bm;
    }
}

template <int MAXN>
__global__ __launch_bounds__(256) void detect_kernel(DetectArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[WIN_H * WIN_STRIDE];
    __shared__ __attribute__((aligned(16))) uint16_t s_hs[WIN_H * HS_STRIDE];
    __shared__ __attribute__((aligned(16))) uint8_t s_nib[2][TH][TW / 4];      // damaged / bright nibbles per 4 pixels
    __shared__ unsigned int s_cnt[2];
    __shared__ unsigned int s_nwork;                                           // groups in the work list (may count past WORK_CAP)
    __shared__ unsigned short s_work[WORK_CAP];                                // (tile row << 5) | group of four columns
    __shared__ double s_rcp[MAXN + 1];
    __shared__ float s_rcpf[MAXN + 1];

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W;
    const int wpr = (W + 63) >> 6;
    // XCD-aware numbering: XCD x (= id % 8) owns tiles [x * per_xcd, (x+1) * per_xcd) of every frame and takes
    // them frame by frame.
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int fz = k / a.per_xcd;                                // plane within this call
    const int tile = xcd * a.per_xcd + (k - fz * a.per_xcd);
    if (fz >= a.count || tile >= a.tiles_x * a.tiles_y) return;
    const int tby = tile / a.tiles_x, tbx = tile - tby * a.tiles_x;
    const int x0 = tbx * TW, y0 = tby * TH;
    const int fi = a.first + fz;            // frame within the chunk
    uint64_t *rawp = a.raw_bits + (size_t)fz * H * wpr;
    uint64_t *brp = a.bright_bits ? a.bright_bits + (size_t)fz * H * wpr : nullptr;

    if (a.keep && !a.keep[fi]) {            // discarded frame: zero planes (workgroup-uniform exit)
        if (tid < 2 * TH) {
            const int ty = tid >> 1, wk = 2 * tbx + (tid & 1);
            if (y0 + ty < H && wk < wpr) {
                rawp[(size_t)(y0 + ty) * wpr + wk] = 0;
                if (brp) brp[(size_t)(y0 + ty) * wpr + wk] = 0;
            }
        }
        return;
    }
    if (tid < 2) s_cnt[tid] = 0;
    if (tid == 2) s_nwork = 0;
    if (tid <= MAXN) { s_rcp[tid] = tid ? 1.0 / (double)tid : 0.0; s_rcpf[tid] = tid ? 1.0f / (float)tid : 0.0f; }

    const uint8_t *frame = a.frames + (size_t)fi * H * W;
    const bool aligned = (W & 3) == 0;

    const int lo = max(0, fi - a.radius), hi = min(a.n, fi + a.radius + 1);
    const int nn = hi - lo - 1;                              // neighbours (centre excluded)
    const int cpos = fi - lo;                                // centre's slot in the window
    const bool full = nn == MAXN;                            // workgroup-uniform
    // FAST: every 4-pixel group of the tile lies inside the image and is dword-aligned (rows below the image are
    // computed from the last row and dropped), nothing but the two planes leaves the kernel
    const bool fast = aligned && x0 + TW <= W && a.bg_out == nullptr;

    // ---- phase A: stage the reflected window, columns x0-16 .. x0+143 ------------------------
    const bool interior = x0 - 16 >= 0 && x0 + TW + 16 <= W && y0 - AX >= 0 && y0 - AX + WIN_H <= H;
    if (interior && (W & 15) == 0 && (reinterpret_cast<uintptr_t>(a.frames) & 15) == 0) {
        const uint8_t *base = frame + (size_t)(y0 - AX) * W + x0 - 16;
        // all of a thread's loads are issued before the first is stored (a rolled loop waits for each load in turn)
        constexpr int NIT = (WIN_H * (WIN_W / 16) + 255) / 256;
        uint4 v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            // (the spare lanes of the last round repeat the last item: an unconditional load keeps v[] in registers)
            const int idx = min(tid + 256 * k, WIN_H * (WIN_W / 16) - 1), r = idx / (WIN_W / 16), q = idx - r * (WIN_W / 16);
            v[k] = *reinterpret_cast<const uint4 *>(base + (size_t)r * W + 16 * q);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int idx = tid + 256 * k, r = idx / (WIN_W / 16), q = idx - r * (WIN_W / 16);
            if (idx < WIN_H * (WIN_W / 16)) *reinterpret_cast<uint4 *>(s_win + r * WIN_STRIDE + 16 * q) = v[k];
        }
    } else {
        const bool interior_x = aligned && x0 - 16 >= 0 && x0 + TW + 16 <= W;
        for (int idx = tid; idx < WIN_H * (WIN_W / 4); idx += 256) {
            const int r = idx / (WIN_W / 4), q = idx - r * (WIN_W / 4);
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
    }
    __syncthreads();

    // ---- phase B: horizontal 30-sums, hs[r][x] = sum_{i<30} win[r][x+1+i] ---------------------
    for (int item = tid; item < WIN_H * (TW / 16); item += 256) {
        const int r = item >> 3, seg = item & 7;
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(s_win + r * WIN_STRIDE) + seg * 4;
        uint32_t d[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) d[j] = rp[j];
        // bytes 1..30: three of d[0], all of d[1..6], three of d[7] (sums of four bytes in one instruction each)
        int s0 = (int)__builtin_amdgcn_sad_u8(d[0] & 0xffffff00u, 0u, 0u);
#pragma unroll
        for (int j = 1; j <= 6; ++j) s0 = (int)__builtin_amdgcn_sad_u8(d[j], 0u, (unsigned)s0);
        s0 = (int)__builtin_amdgcn_sad_u8(d[7] & 0x00ffffffu, 0u, (unsigned)s0);
        int sv[16];
        sv[0] = s0;
        sv[1] = hslide<1>(sv[0], d);   sv[2] = hslide<2>(sv[1], d);   sv[3] = hslide<3>(sv[2], d);   sv[4] = hslide<4>(sv[3], d);
        sv[5] = hslide<5>(sv[4], d);   sv[6] = hslide<6>(sv[5], d);   sv[7] = hslide<7>(sv[6], d);   sv[8] = hslide<8>(sv[7], d);
        sv[9] = hslide<9>(sv[8], d);   sv[10] = hslide<10>(sv[9], d); sv[11] = hslide<11>(sv[10], d); sv[12] = hslide<12>(sv[11], d);
        sv[13] = hslide<13>(sv[12], d); sv[14] = hslide<14>(sv[13], d); sv[15] = hslide<15>(sv[14], d);
        uint32_t out[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) out[t] = (uint32_t)sv[2 * t] | ((uint32_t)sv[2 * t + 1] << 16);
        uint32_t *hp = reinterpret_cast<uint32_t *>(s_hs + r * HS_STRIDE + seg * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[j] = out[j];
    }
    __syncthreads();

    // ---- phase C: per-pixel tests; a thread owns columns 4*cg .. 4*cg+3 of rows 8*rg .. 8*rg+7 and slides the
    // vertical 30-sum of its columns down those rows in registers ---------------------------------------------
    unsigned int c_lt = 0, c_gt = 0;
    if (full && fast) pixel_rows<MAXN, true, true>(a, s_win, s_hs, s_nib, s_rcp, s_rcpf, &s_nwork, s_work, x0, y0, fi, lo, nn, cpos, c_lt, c_gt);
    else if (full) pixel_rows<MAXN, true, false>(a, s_win, s_hs, s_nib, s_rcp, s_rcpf, &s_nwork, s_work, x0, y0, fi, lo, nn, cpos, c_lt, c_gt);
    else if (fast) pixel_rows<MAXN, false, true>(a, s_win, s_hs, s_nib, s_rcp, s_rcpf, &s_nwork, s_work, x0, y0, fi, lo, nn, cpos, c_lt, c_gt);
    else pixel_rows<MAXN, false, false>(a, s_win, s_hs, s_nib, s_rcp, s_rcpf, &s_nwork, s_work, x0, y0, fi, lo, nn, cpos, c_lt, c_gt);
    if (a.dense) {
        // wave-level reduce, then one LDS atomic per wave
        for (int off = 32; off > 0; off >>= 1) {
            c_lt += __shfl_down(c_lt, off);
            c_gt += __shfl_down(c_gt, off);
        }
        if ((tid & 63) == 0) { atomicAdd(&s_cnt[0], c_lt); atomicAdd(&s_cnt[1], c_gt); }
    }
    __syncthreads();

    // ---- phase D: 16 nibbles -> one 64-bit word; thread = (plane, row, word of the tile) -------------------
    if (tid < 4 * TH) {
        const int plane = tid / (2 * TH), ty = (tid >> 1) & (TH - 1), half = tid & 1;
        const uint4 q = *reinterpret_cast<const uint4 *>(&s_nib[plane][ty][16 * half]);
        const uint64_t word = (uint64_t)(nibbles4(q.x) | (nibbles4(q.y) << 16)) |
                              ((uint64_t)(nibbles4(q.z) | (nibbles4(q.w) << 16)) << 32);
        const int wk = 2 * tbx + half;
        uint64_t *dst = plane ? brp : rawp;
        if (dst && y0 + ty < H && wk < wpr) dst[(size_t)(y0 + ty) * wpr + wk] = word;
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
    a.n = n; a.H = H; a.W = W; a.first = first; a.radius = ctx->p.sliding_window_radius; a.count = count;
    a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TH - 1) / TH;
    a.per_xcd = (a.tiles_x * a.tiles_y + 7) / 8;
    a.frame_gt = d.frame_gt; a.est_lt = d.est_lt; a.bg_gt = d.bg_gt; a.coeff = ctx->p.pixel_std_coeff;
    if (dense_counts)
        IPX_CHECK_HIP(hipMemsetAsync(dense_counts, 0, sizeof(int64_t) * 2 * (size_t)count, ctx->stream));
    const size_t blocks = (size_t)8 * a.per_xcd * count;
    if (blocks > 0x7fffffffu) {
        ipx_set_error("ipx_detect_raw: %d planes of %d x %d are too many workgroups for one launch", count, W, H);
        return IPX_ERR_UNSUPPORTED;
    }
    dim3 grid((unsigned)blocks), block(256);
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

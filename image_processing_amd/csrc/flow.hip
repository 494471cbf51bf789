// flow.hip -- the frame-difference motion test: dense Farneback flow between consecutive frames
// and its mean magnitude (gfx950).
//
// Replaces compute_optical_flow_metric, ip.py:852-873:
//   cv2.calcOpticalFlowFarneback(prev, curr, None, 0.5, 3, 15, 3, 5, 2, 0)   ip.py:858-867
//   np.mean(cv2.cartToPolar(fx, fy)[0])                                      ip.py:868
// and the discard rule `flow <= flow_threshold` of ip.py:559.
//
// The arithmetic follows the published OpenCV algorithm as SURVEY.md App. A.6 specifies it
// (pyramid = Gaussian blur of the full-resolution frame + linear resize; polynomial expansion
// with 11-tap separable filters, vertical pass in f32, horizontal pass in f64; matrix update with
// a bilinear sample of the second expansion and border damping; 15x15 box window in f64 and a
// regularised 2x2 solve).  Every per-pixel expression is evaluated with the same types and
// operation order (no FMA contraction), and so are the vertical running sums of the 15x15 window,
// f32-rounded updates included.  Two reductions are formed in a different order: the window's
// horizontal 15-sums (f64; a flow value can differ from the reference's by one f32 ulp, rarely --
// none of the ~1e9 values of the parity tests does) and the mean magnitude (f64 tree instead of
// numpy's f32 pairwise sum; inside the 1e-5 relative tolerance of the float metric).
//
// MI355X design: a batch of up to 512 consecutive pairs is processed per launch so that the coarse pyramid
// levels still fill 256 CUs; the polynomial expansion of a frame is computed once and serves both pairs it
// belongs to (computing the expansions of batch b+1 on a side stream under the iterations of batch b was
// tried: the kernels overlap but slow each other down by the same amount); the update matrices live only in
// LDS (flow_iter_kernel); upsampling between levels and the
// magnitude reduction are variants of the iteration kernel; workgroup ids are mapped so that the pairs that
// share an expansion run on the same XCD (one L2).
#include "ipx_internal.h"
#include "flow_iter.h"
#include <math.h>
#include <float.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <type_traits>
#include <utility>

namespace {

constexpr int MAX_LEVELS = 8;
constexpr int POLY_N = 5;
constexpr int MAX_KS = 31;

struct LevelGeom {
    int lh, lw;          // level size
    int ksize;           // Gaussian aperture (odd)
    int mode;            // 0 = same size (copy), 1 = exact 2x2 area average, 2 = two-tap linear
    float kern[MAX_KS];  // Gaussian taps, f32
    // device tables (inside SLOT_FLOW_TAB)
    const int *x_ofs;    // [lw]  source column of the first tap
    const float *x_a;    // [lw]  weight of the second tap
    const int *y_ofs;    // [lh][2] clamped source rows
    const float *y_b;    // [lh]  weight of the second row
    // tiled pyramid kernel: the two source columns / rows every level pixel samples (equal in mode 0)
    const int *xs2;      // [lw][2]
    const int *ys2;      // [lh][2]
    int exact_scale;     // S when W = S lw, H = S lh, S = 4 or 8 and every resize tap is (S d + S/2 - 1, weight 0.5): pyr_exact_kernel
    int tox, toy;        // level pixels per tile
    int src_rows, src_cols, src_pitch, tmp_pitch;   // staged source window (max over tiles) and LDS pitches
    unsigned lds_bytes;
    // upsampling of the previous (coarser) level's flow to this level
    const int *ux_ofs; const float *ux_a; const int *uy_ofs; const float *uy_b;
};

struct PolyConst {
    float g[POLY_N + 1], xg[POLY_N + 1], xxg[POLY_N + 1];
    double gd[POLY_N + 1], xxgd[POLY_N + 1];             // g and xxg widened once on the host
    double ig11, ig03, ig33, ig55;
};

}  // namespace

struct ipx_flow_plan {
    int H, W, levels;               // levels = index of the coarsest level (0 = full resolution only)
    LevelGeom lv[MAX_LEVELS];
    PolyConst pc;
};

void ipx_flow_plan_free(ipx_flow_plan *plan) { delete plan; }

namespace {

inline int cv_round(double v) { return (int)lrint(v); }
inline int cv_floor(double v) { return (int)floor(v); }

// getGaussianKernel(n, sigma, CV_32F)
void gaussian_kernel(int n, double sigma, float *k)
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
    double tmp[MAX_KS], sum = 0.0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        tmp[i] = exp(scale2 * x * x);
        sum += tmp[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(tmp[i] * sum);
}

// FarnebackPrepareGaussian: taps and the inverse Gram entries (closed-form block inverse)
void poly_prepare(int n, double sigma, PolyConst &pc)
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    float gb[2 * POLY_N + 1], xgb[2 * POLY_N + 1], xxgb[2 * POLY_N + 1];
    float *g = gb + n, *xg = xgb + n, *xxg = xxgb + n;
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
    const double a = G00, b = G11, c = G33, d = G55;
    const double det = a * (c * c - d * d) - 2.0 * b * b * (c - d);
    pc.ig11 = 1.0 / G11;
    pc.ig03 = (b * d - b * c) / det;
    pc.ig33 = (a * c - b * b) / det;
    pc.ig55 = 1.0 / G55;
    for (int k = 0; k <= n; ++k) { pc.g[k] = g[k]; pc.xg[k] = xg[k]; pc.xxg[k] = xxg[k]; pc.gd[k] = (double)g[k]; pc.xxgd[k] = (double)xxg[k]; }
}

// cv::resize(INTER_LINEAR) coordinate tables for one axis: dst length dn from src length sn.
void linear_table_x(int sn, int dn, std::vector<int> &ofs, std::vector<float> &a)
{
    const double scale = 1.0 / ((double)dn / sn);
    ofs.resize(dn); a.resize(dn);
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= sn - 1) { f = 0; s = sn - 1; }
        ofs[d] = s; a[d] = f;
    }
}
void linear_table_y(int sn, int dn, std::vector<int> &ofs2, std::vector<float> &b)
{
    const double scale = 1.0 / ((double)dn / sn);
    ofs2.resize(2 * (size_t)dn); b.resize(dn);
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = cv_floor(f);
        f -= s;
        ofs2[2 * d] = s < 0 ? 0 : (s > sn - 1 ? sn - 1 : s);
        ofs2[2 * d + 1] = s + 1 < 0 ? 0 : (s + 1 > sn - 1 ? sn - 1 : s + 1);
        b[d] = f;
    }
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while ((unsigned)p >= (unsigned)len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

struct KernTaps { float k[MAX_KS]; };

// Tap order of GaussianBlur(frame as f32, (KS,KS), sigma), BORDER_REFLECT_101, as the separable filter
// engine evaluates it: row pass then column pass in f32; symmetric pairing for apertures <= 5 in the
// row pass, left-to-right otherwise; symmetric pairing from the centre outwards in the column pass.
struct PyrArgs {
    const uint8_t *frames;     // frame f of this launch = frames + frame_index(f) * H * W
    const uint8_t *extra;      // optional frame standing in for index -1 (left halo), may be null
    int first;                 // chunk index of launch frame 0 (may be -1 when `extra` is given)
    int H, W, lh, lw, mode;
    int tox, toy, src_pitch, tmp_pitch, src_rows;
    int tiles_x, tiles_y, nf;
    const int *xs2; const float *x_a; const int *ys2; const float *y_b;
    float *out;                // [nf][lh][lw]
    KernTaps kt;
};

// Pyramid level image of one frame: GaussianBlur(frame as f32, KS x KS) sampled at the source
// positions the linear resize needs, then the resize itself.  One block = tox x toy level pixels:
//   1. the source window (reflect-101 applied) is staged in LDS as f32,
//   2. the row pass is evaluated for every staged row at the sampled columns only (lanes = rows,
//      odd pitches: conflict-free), tap order as blurred_at/row_pass above,
//   3. each level pixel runs the column pass at its (row, column) samples and combines them
//      (copy / 2x2 mean / two-tap linear, cv::resize INTER_LINEAR semantics).
template <int KS>
__global__ __launch_bounds__(256) void pyr_level_kernel(PyrArgs a)
{
    extern __shared__ __align__(16) unsigned char pyr_smem[];
    constexpr int R = KS / 2;
    float *s_src = reinterpret_cast<float *>(pyr_smem);
    float *s_tmp = s_src + a.src_rows * a.src_pitch;
    int *s_col = reinterpret_cast<int *>(s_tmp + a.src_rows * a.tmp_pitch);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ns = a.mode == 0 ? 1 : 2;
    // tiles are numbered frame-major and dealt to the 8 XCDs in contiguous ranges: neighbouring tiles, which
    // share their source halo, run on the same L2
    const int per_frame = a.tiles_x * a.tiles_y, total = per_frame * a.nf, per_xcd = (total + 7) >> 3;
    const int tk = (int)(blockIdx.x >> 3), tile = (int)(blockIdx.x & 7) * per_xcd + tk;
    if (tk >= per_xcd || tile >= total) return;
    const int f = tile / per_frame, trem = tile - f * per_frame;
    const int tby = trem / a.tiles_x, tbx = trem - tby * a.tiles_x;
    const int dx0 = tbx * a.tox, dy0 = tby * a.toy;
    const int ndx = min(a.tox, a.lw - dx0), ndy = min(a.toy, a.lh - dy0);
    const int fi = a.first + f;
    const uint8_t *frame = fi < 0 ? a.extra : a.frames + (size_t)fi * a.H * a.W;
    const int xlo = a.xs2[2 * dx0], xhi = a.xs2[2 * (dx0 + ndx - 1) + 1];
    const int ylo = a.ys2[2 * dy0], yhi = a.ys2[2 * (dy0 + ndy - 1) + 1];
    const int c_lo = xlo - R, c_hi = xhi + R;              // staged columns c_org .. c_hi, c_org = c_lo rounded down to 4
    const int c_org = c_lo & ~3, ndw = (c_hi - c_org + 4) >> 2;
    const int r_lo = ylo - R, nrows = yhi - ylo + 1 + 2 * R;
    const int nsamp = ns * ndx;
    // 1. stage the source window as f32 (LDS column = source column - c_org)
    if (c_org >= 0 && c_org + 4 * ndw <= a.W && r_lo >= 0 && r_lo + nrows <= a.H && (a.W & 3) == 0) {
        // interior tile: aligned dword loads, four independent loads in flight per thread
        const int total = nrows * ndw;
        const int dq = 256 % ndw, dr = 256 / ndw;
        int r = tid / ndw, q = tid - r * ndw;
        const uint8_t *base = frame + (size_t)r_lo * a.W + c_org;
        for (int i0 = tid; i0 < total; i0 += 4 * 256) {
            uint32_t v[4]; int rr[4], qq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rr[u] = r; qq[u] = q;
                v[u] = (i0 + u * 256 < total) ? *reinterpret_cast<const uint32_t *>(base + (size_t)r * a.W + 4 * q) : 0u;
                q += dq; r += dr;
                if (q >= ndw) { q -= ndw; ++r; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < total) {
                    float *d = s_src + rr[u] * a.src_pitch + 4 * qq[u];
                    d[0] = (float)(v[u] & 0xff); d[1] = (float)((v[u] >> 8) & 0xff);
                    d[2] = (float)((v[u] >> 16) & 0xff); d[3] = (float)(v[u] >> 24);
                }
        }
    } else {
        const int ncols = c_hi - c_org + 1;
        for (int r = wv; r < nrows; r += 4) {
            const uint8_t *row = frame + (size_t)reflect101(r_lo + r, a.H) * a.W;
            for (int c = lane; c < ncols; c += 64) s_src[r * a.src_pitch + c] = (float)row[reflect101(c_org + c, a.W)];
        }
    }
    for (int s = tid; s < nsamp; s += 256)
        s_col[s] = (ns == 1 ? a.xs2[2 * (dx0 + s)] : a.xs2[2 * (dx0 + (s >> 1)) + (s & 1)]) - c_org;
    __syncthreads();
    // 2. row pass: wave -> sample, lane -> staged row.  The two samples of a level pixel are usually
    // neighbouring columns: they are then evaluated from one set of KS+1 LDS reads (same taps, same order).
    const bool paired = KS > 5 && ns == 2;
    for (int sp = wv; paired && sp < ndx; sp += 4) {
        const int c0 = s_col[2 * sp], c1 = s_col[2 * sp + 1];
        if (c1 != c0 + 1) continue;
        for (int r = lane; r < nrows; r += 64) {
            const float *S = s_src + r * a.src_pitch + c0 - R;
            float x[KS + 1];
#pragma unroll
            for (int i = 0; i <= KS; ++i) x[i] = S[i];
            float v0 = x[0] * a.kt.k[0], v1 = x[1] * a.kt.k[0];
#pragma unroll
            for (int i = 1; i < KS; ++i) { v0 += x[i] * a.kt.k[i]; v1 += x[i + 1] * a.kt.k[i]; }
            s_tmp[r * a.tmp_pitch + 2 * sp] = v0;
            s_tmp[r * a.tmp_pitch + 2 * sp + 1] = v1;
        }
    }
    for (int s = wv; s < nsamp; s += 4) {
        const int cc = s_col[s];
        if (paired && s_col[s ^ 1] == cc + 1 - 2 * (s & 1)) continue;      // done above
        for (int r = lane; r < nrows; r += 64) {
            const float *S = s_src + r * a.src_pitch + cc;
            float v;
            if (KS <= 5) {
                v = S[0] * a.kt.k[R];
#pragma unroll
                for (int j = 1; j <= R; ++j) v += (S[-j] + S[j]) * a.kt.k[R + j];
            } else {
                v = S[-R] * a.kt.k[0];
#pragma unroll
                for (int i = 1; i < KS; ++i) v += S[i - R] * a.kt.k[i];
            }
            s_tmp[r * a.tmp_pitch + s] = v;
        }
    }
    __syncthreads();
    // 3. column pass + resize
    auto col_pass = [&](int r, int s) -> float {
        const float *T = s_tmp + r * a.tmp_pitch + s;
        float v = a.kt.k[R] * T[0];
#pragma unroll
        for (int j = 1; j <= R; ++j) v += a.kt.k[R + j] * (T[j * a.tmp_pitch] + T[-j * a.tmp_pitch]);
        return v;
    };
    // the two row samples of a level pixel are usually neighbouring rows: both column passes from KS+1 reads
    auto col_pass2 = [&](int ra, int rb, int s, float &va, float &vb) {
        if (KS <= 5 || rb != ra + 1) { va = col_pass(ra, s); vb = col_pass(rb, s); return; }
        const float *T = s_tmp + (ra - R) * a.tmp_pitch + s;
        float t[KS + 1];
#pragma unroll
        for (int i = 0; i <= KS; ++i) t[i] = T[i * a.tmp_pitch];
        va = a.kt.k[R] * t[R]; vb = a.kt.k[R] * t[R + 1];
#pragma unroll
        for (int j = 1; j <= R; ++j) {
            va += a.kt.k[R + j] * (t[R + j] + t[R - j]);
            vb += a.kt.k[R + j] * (t[R + 1 + j] + t[R + 1 - j]);
        }
    };
    const int tox_sh = 31 - __builtin_clz(a.tox);
    for (int idx = tid; idx < a.tox * ndy; idx += 256) {
        const int dyl = idx >> tox_sh, dxl = idx - (dyl << tox_sh);   // tox is a power of two (build_plan)
        if (dxl >= ndx) continue;
        const int dx = dx0 + dxl, dy = dy0 + dyl;
        const int ra = a.ys2[2 * dy] - r_lo, rb = a.ys2[2 * dy + 1] - r_lo;
        float v;
        if (a.mode == 0) {
            v = col_pass(ra, dxl);
        } else if (a.mode == 1) {
            float s00, s01, s10, s11;
            col_pass2(ra, rb, 2 * dxl, s00, s10);
            col_pass2(ra, rb, 2 * dxl + 1, s01, s11);
            v = (s00 + s01 + s10 + s11) * 0.25f;
        } else {
            const float a1 = a.x_a[dx], a0 = 1.f - a1, b1 = a.y_b[dy], b0 = 1.f - b1;
            float h0, h1;
            col_pass2(ra, rb, 2 * dxl, h0, h1);
            if (a.xs2[2 * dx] < a.W - 1) {                    // (right edge: single tap)
                float g0, g1;
                col_pass2(ra, rb, 2 * dxl + 1, g0, g1);
                h0 = h0 * a0 + g0 * a1;
                h1 = h1 * a0 + g1 * a1;
            }
            v = h0 * b0 + h1 * b1;
        }
        a.out[((size_t)f * a.lh + dy) * a.lw + dx] = v;
    }
}

// The same level image for the geometry the usual pyramid has at levels 2 and 3: W = S * lw and H = S * lh with S = 4
// or 8, so that every level pixel is the mean of the blurred image at source columns S dx + S/2 - 1, S dx + S/2 and
// rows S dy + S/2 - 1, S dy + S/2 (both linear weights are exactly 0.5; launch_level_expansion checks the tables).
// pyr_level_kernel is generic over scales (run-time tables, lanes = rows, which caps a tile at 64 staged rows and, with
// its f32 window, at 8 level columns for S = 4): 815 vector instructions per level pixel at S = 4 where the arithmetic
// needs 200.  Here the tile is TOX x TOY level pixels and nothing is looked up:
//   1. the source window goes to LDS as f32, one aligned quad of columns per item (reflect-101 outside the frame);
//   2. row pass: item = (window row, level column), lanes along the columns: the KS + 1 window values of the column
//      pair are read as aligned 16-byte quads (conflict-free for S = 4), taps left to right as in pyr_level_kernel;
//   3. column pass + mean: thread = level pixel, the KS + 1 rows of both columns as 8-byte reads, taps paired from the
//      centre outwards, then (h0 a0 + g0 a1) b0 + (h1 a0 + g1 a1) b1 with a0 = a1 = b0 = b1 = 0.5.
// Every expression has the operations and the order of pyr_level_kernel<KS> in mode 2; the parity tests compare the
// level images of both kernels through the flow fields, which stay bit-identical.
typedef float pyr_f32x4 __attribute__((ext_vector_type(4)));
typedef float pyr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void asm_keep(pyr_f32x2 &v) { asm volatile("" : "+v"(v)); }
template <int OFF>
__device__ __forceinline__ void pyr_read_b64(unsigned addr, pyr_f32x2 &v)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
// t[k] = the 8 bytes at addr + k * PITCH for k = 0 .. N-1, waited for
template <int PITCH, int... K>
__device__ __forceinline__ void pyr_read_rows(unsigned addr, pyr_f32x2 (&t)[sizeof...(K)], std::integer_sequence<int, K...>)
{
    (pyr_read_b64<K * PITCH>(addr, t[K]), ...);
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
    (asm_keep(t[K]), ...);
}
template <int KS, int S, int TOX, int TOY>
__global__ __launch_bounds__(256) void pyr_exact_kernel(PyrArgs a)
{
    static_assert((KS == 9 && S == 4) || (KS == 19 && S == 8), "the row pass reads 3 or 6 quads");
    constexpr int R = KS / 2;
    constexpr int OFF = (R - S / 2 + 1 + 3) & ~3;                 // window column 0 = source column S dx0 - OFF
    constexpr int CWN = S * TOX + ((R - S / 2 + OFF + 1 + 3) & ~3); // staged columns (a multiple of 4)
    constexpr int CW = S == 8 ? CWN + 4 : CWN;                     // pitch: with 8 lanes a row, 32 bytes apart, 84 floats leave the
                                                                   // 16-byte reads of the row pass two-way conflicts, 80 four-way
    constexpr int NR = S * (TOY - 1) + 2 + 2 * R;                  // staged rows, first = source row S dy0 + S/2 - 1 - R
    constexpr int Q0 = S / 2 - 1 - R + OFF;                        // first window value of a column pair, relative to S dxl
    constexpr int NQ = (Q0 + KS + 1 + 3) / 4;                      // quads read per row-pass item
    constexpr int TP = 2 * TOX;                                    // row-pass results per window row
    static_assert(Q0 >= 0 && S * (TOX - 1) + 4 * NQ <= CW, "row-pass reads stay inside the window");
    __shared__ __align__(16) float s_src[NR * CW];
    __shared__ __align__(16) float s_tmp[NR * TP];
    const int tid = threadIdx.x;
    const int per_frame = a.tiles_x * a.tiles_y, total = per_frame * a.nf, per_xcd = (total + 7) >> 3;
    const int tk = (int)(blockIdx.x >> 3), tile = (int)(blockIdx.x & 7) * per_xcd + tk;
    if (tk >= per_xcd || tile >= total) return;
    const int f = tile / per_frame, trem = tile - f * per_frame;
    const int tby = trem / a.tiles_x, tbx = trem - tby * a.tiles_x;
    const int dx0 = tbx * TOX, dy0 = tby * TOY;
    const int fi = a.first + f;
    const uint8_t *frame = fi < 0 ? a.extra : a.frames + (size_t)fi * a.H * a.W;
    const int c_org = S * dx0 - OFF, r_lo = S * dy0 + S / 2 - 1 - R;
    // 1. window.  (One reflection suffices: a level of this path has at least 32 pixels a side, so the frame has 32 S,
    // and the window reaches at most R + S / 2 beyond it.  Item -> (row, quad) by stepping, not by division.)
    {
        constexpr int QW = CWN / 4, DR = 256 / QW, DQ = 256 % QW;
        const bool inside = c_org >= 0 && c_org + CWN <= a.W;      // (W is a multiple of 4 on this path)
        auto refl = [](int p, int len) { return p < 0 ? -p : (p > len - 1 ? 2 * (len - 1) - p : p); };
        constexpr int NIT = (NR * QW + 255) / 256;
        int r = tid / QW, q = tid - r * QW;
        if (inside) {
            // all of a thread's loads are issued before the first is converted (rolled up, every round would wait
            // for its own trip to memory)
            uint32_t d[NIT]; int rr[NIT], qq[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                rr[k] = r; qq[k] = q;
                const int rc = min(r, NR - 1);
                d[k] = *reinterpret_cast<const uint32_t *>(frame + (size_t)refl(r_lo + rc, a.H) * a.W + c_org + 4 * q);
                q += DQ; r += DR;
                if (q >= QW) { q -= QW; ++r; }
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k)
                if (rr[k] < NR)
                    *reinterpret_cast<float4 *>(s_src + rr[k] * CW + 4 * qq[k]) =
                        make_float4((float)(d[k] & 0xff), (float)((d[k] >> 8) & 0xff), (float)((d[k] >> 16) & 0xff), (float)(d[k] >> 24));
        } else {
            while (r < NR) {
                const uint8_t *row = frame + (size_t)refl(r_lo + r, a.H) * a.W;
                const int c = c_org + 4 * q;
                *reinterpret_cast<float4 *>(s_src + r * CW + 4 * q) =
                    make_float4((float)row[refl(c, a.W)], (float)row[refl(c + 1, a.W)], (float)row[refl(c + 2, a.W)], (float)row[refl(c + 3, a.W)]);
                q += DQ; r += DR;
                if (q >= QW) { q -= QW; ++r; }
            }
        }
    }
    __syncthreads();
    // 2. row pass
    for (int i = tid; i < NR * TOX; i += 256) {
        const int r = i / TOX, dxl = i - r * TOX;
        // (ds_read_b128 by hand: the compiler splits a float4 read into 4- and 8-byte reads at odd offsets so that the
        // pairs it multiplies land in aligned registers, and with the lanes 16 or 32 bytes apart those run into
        // four-way bank conflicts -- the LDS was 91 % busy, 58 % of it conflicts)
        const unsigned sa = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)(s_src + r * CW + S * dxl);
        pyr_f32x4 xq[6];
        asm volatile("ds_read_b128 %0, %1" : "=v"(xq[0]) : "v"(sa) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(xq[1]) : "v"(sa) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(xq[2]) : "v"(sa) : "memory");
        if (NQ > 3) {
            asm volatile("ds_read_b128 %0, %1 offset:48" : "=v"(xq[3]) : "v"(sa) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(xq[4]) : "v"(sa) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:80" : "=v"(xq[5]) : "v"(sa) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]), "+v"(xq[4]), "+v"(xq[5]) : : "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]) : : "memory");
        }
        float x[4 * NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { x[4 * q] = xq[q].x; x[4 * q + 1] = xq[q].y; x[4 * q + 2] = xq[q].z; x[4 * q + 3] = xq[q].w; }
        float v0 = x[Q0] * a.kt.k[0], v1 = x[Q0 + 1] * a.kt.k[0];
#pragma unroll
        for (int k = 1; k < KS; ++k) { v0 += x[Q0 + k] * a.kt.k[k]; v1 += x[Q0 + k + 1] * a.kt.k[k]; }
        *reinterpret_cast<float2 *>(s_tmp + r * TP + 2 * dxl) = make_float2(v0, v1);
    }
    __syncthreads();
    // 3. column pass + mean
    for (int i = tid; i < TOX * TOY; i += 256) {
        const int dyl = i / TOX, dxl = i - dyl * TOX;
        const int dx = dx0 + dxl, dy = dy0 + dyl;
        if (dx >= a.lw || dy >= a.lh) continue;
        // (single ds_read_b64: paired into ds_read2_b64 they take the LDS array twice as long)
        const unsigned ta = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float *)(s_tmp + (S * dyl) * TP + 2 * dxl);
        pyr_f32x2 t[KS + 1];
        pyr_read_rows<TP * 4>(ta, t, std::make_integer_sequence<int, KS + 1>());
        float h0 = a.kt.k[R] * t[R].x, h1 = a.kt.k[R] * t[R + 1].x, g0 = a.kt.k[R] * t[R].y, g1 = a.kt.k[R] * t[R + 1].y;
#pragma unroll
        for (int j = 1; j <= R; ++j) {
            h0 += a.kt.k[R + j] * (t[R + j].x + t[R - j].x);
            h1 += a.kt.k[R + j] * (t[R + 1 + j].x + t[R + 1 - j].x);
            g0 += a.kt.k[R + j] * (t[R + j].y + t[R - j].y);
            g1 += a.kt.k[R + j] * (t[R + 1 + j].y + t[R + 1 - j].y);
        }
        const float a1 = 0.5f, a0 = 1.f - a1, b1 = 0.5f, b0 = 1.f - b1;
        h0 = h0 * a0 + g0 * a1;
        h1 = h1 * a0 + g1 * a1;
        a.out[((size_t)f * a.lh + dy) * a.lw + dx] = h0 * b0 + h1 * b1;
    }
}

// FarnebackPolyExp: I [nf][h][w] -> R [nf][{float4 [h][w], float [h][w]}].  Workgroup = 64 columns x 16 rows:
//   1. the (16+10) x (64+10) window of I (rows and columns clamped: the reference replicates them) goes to LDS,
//   2. the vertical pass (f32, taps paired from the centre outwards) of the 74 columns goes to LDS,
//   3. the horizontal pass accumulates in f64.  b1 and b4 are sums of products of two f32 values held in
//      f64: the products are exact, so the fused multiply-add rounds exactly like the reference's mul + add.
// Tiles are numbered frame-major and dealt to the 8 XCDs in contiguous ranges, so that vertically adjacent
// tiles (which share 10 of 26 window rows) run on the same L2.
// Measured: the kernel is VALU-bound (~200 instructions per pixel, HBM traffic = the algorithmic bytes).
// Putting two rows / two columns into packed f32 instructions (v_pk_add_f32, v_pk_mul_f32) cut the
// instruction count by a quarter and made it 12 % slower, so the scalar form stays.
// Levels 0 and 1 of the usual pyramid (3x3 Gaussian, then a copy or an exact 2x2 mean) when H and W are
// multiples of 4: no LDS, one thread = a 4x4 block of blurred full-resolution pixels from its 6x6 source
// window (three loads per source row, neighbours hit L1), same operation order as pyr_level_kernel<3>:
// row pass S[0]*k1 + (S[-1]+S[1])*k2, column pass k1*T[0] + k2*(T[1]+T[-1]), mean (s00+s01+s10+s11)*0.25.
template <int MODE>
__global__ __launch_bounds__(256) void pyr3_kernel(PyrArgs a)
{
    const int bx = 4 * (blockIdx.x * 64 + (threadIdx.x & 63)), by = 4 * (blockIdx.y * 4 + (threadIdx.x >> 6));
    if (bx >= a.W || by >= a.H) return;
    const int f = blockIdx.z, fi = a.first + f;
    const uint8_t *frame = fi < 0 ? a.extra : a.frames + (size_t)fi * a.H * a.W;
    const float k1 = a.kt.k[1], k2 = a.kt.k[2];
    const int lane = threadIdx.x & 63;
    float rp[6][4];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const uint8_t *row = frame + (size_t)reflect101(by - 1 + r, a.H) * a.W;
        const uint32_t mid = *reinterpret_cast<const uint32_t *>(row + bx);
        // the two pixels beside the dword are in the neighbouring lanes' dwords (ds_bpermute: the kernel has no other
        // LDS traffic); only the first and the last lane of a wave load theirs (18 full loads per thread before)
        const uint32_t lmid = (uint32_t)__shfl_up((int)mid, 1), rmid = (uint32_t)__shfl_down((int)mid, 1);
        float p[6];
        p[0] = (float)(bx > 0 ? (lane > 0 ? (int)(lmid >> 24) : (int)row[bx - 1]) : (int)((mid >> 8) & 0xff));   // reflect-101: column -1 is column 1
        p[1] = (float)(mid & 0xff); p[2] = (float)((mid >> 8) & 0xff);
        p[3] = (float)((mid >> 16) & 0xff); p[4] = (float)(mid >> 24);
        p[5] = (float)(bx + 4 < a.W ? (lane < 63 ? (int)(rmid & 0xff) : (int)row[bx + 4]) : (int)((mid >> 16) & 0xff));   // column W is column W-2
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = p[c + 1] * k1;
            v += (p[c] + p[c + 2]) * k2;
            rp[r][c] = v;
        }
    }
    float bl[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = k1 * rp[r + 1][c];
            v += k2 * (rp[r + 2][c] + rp[r][c]);
            bl[r][c] = v;
        }
    if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4 *>(a.out + ((size_t)f * a.lh + by + r) * a.lw + bx) = make_float4(bl[r][0], bl[r][1], bl[r][2], bl[r][3]);
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float v0 = (bl[2 * r][0] + bl[2 * r][1] + bl[2 * r + 1][0] + bl[2 * r + 1][1]) * 0.25f;
            const float v1 = (bl[2 * r][2] + bl[2 * r][3] + bl[2 * r + 1][2] + bl[2 * r + 1][3]) * 0.25f;
            *reinterpret_cast<float2 *>(a.out + ((size_t)f * a.lh + (by >> 1) + r) * a.lw + (bx >> 1)) = make_float2(v0, v1);
        }
    }
}


constexpr int PX_TW = 64, PX_TH = 16, PX_SW = PX_TW + 2 * POLY_N, PX_SH = PX_TH + 2 * POLY_N, PX_SP = PX_SW + 1;
constexpr int PX_IP = 80, PX_OX = 3;      // staged window: 80 columns from x0 - 8 (16-byte aligned rows); window column c is staged column c + 3
// FUSED: level 0 of the usual pyramid (the frame itself under a 3x3 Gaussian, pyr3_kernel<0>) is evaluated while the
// window is staged, straight from the uint8 frame, instead of being written to HBM by one kernel and read back by
// this one: row pass p*k1 + (left + right)*k2 of the 28 source rows the window touches (reflect-101 at the image
// borders, as the blur has them) into the LDS the vertical pass uses later, then the column pass
// k1*mid + k2*(below + above) at the window's (replicated) rows -- the operations and their order are pyr3_kernel's.
struct PxSrc { const uint8_t *frames, *extra; int first; float k1, k2; };
template <bool FUSED>
__global__ __launch_bounds__(256) void polyexp_kernel(const float *I, int h, int w, int tiles_x, int tiles_y, int nf, PolyConst pc, float *R, PxSrc src)
{
    __shared__ __align__(16) float s_in[PX_SH * PX_IP];
    __shared__ __align__(16) float s_t[PX_TH][3][PX_SP];   // [row][t0 | t1 | t2][column]: the three planes of a row within one read offset range
    const int tid = threadIdx.x;
    const int per_frame = tiles_x * tiles_y, total = per_frame * nf, per_xcd = (total + 7) >> 3;
    const int k = (int)(blockIdx.x >> 3);
    const int tile = (int)(blockIdx.x & 7) * per_xcd + k;
    if (k >= per_xcd || tile >= total) return;
    const int f = tile / per_frame, rem = tile - f * per_frame;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int x0 = tx * PX_TW, y0 = ty * PX_TH;
    const float *img = I + (size_t)f * h * w;
    // 1. window.  All of a thread's loads are issued before the first is stored (a rolled loop waits for every load
    // before it issues the next).  Tiles away from the left and right image borders take whole 16-byte groups (three
    // loads per thread); the others clamp every column.
    if (FUSED) {
        constexpr int ER = PX_SH + 2, NQ = PX_IP / 4;          // source rows y0 - 6 .. y0 + 21, reflected: row-pass row e
        float *s_rp = &s_t[0][0][0];                            // [ER][PX_IP], free until the vertical pass
        static_assert(ER * PX_IP <= 3 * PX_TH * PX_SP, "row-pass rows must fit into s_t");
        const int fi = src.first + f;
        const uint8_t *frame = fi < 0 ? src.extra : src.frames + (size_t)fi * h * w;
        const float k1 = src.k1, k2 = src.k2;
        // (thread -> (row, quad) once: quad q = tid % 20 of rows e0, e0 + 12, e0 + 24; threads 240-255 idle here)
        constexpr int QROWS = 240 / NQ;
        static_assert(NQ * QROWS == 240, "20 quads x 12 rows per round");
        const int sq = tid % NQ, se0 = tid / NQ;
        const bool sact = tid < NQ * QROWS;
        // rows y0 - 6 .. y0 + 21 inside the image: no reflection in the row pass, no replication in the column pass
        const bool yin = y0 >= POLY_N + 1 && y0 + PX_TH + POLY_N <= h - 1;
        if (x0 >= 8 && x0 + PX_IP - 8 <= w) {                   // (w is a multiple of 4 on this path)
            constexpr int NIT = (ER + QROWS - 1) / QROWS;
            static_assert(NIT == 3 && 2 * QROWS <= ER - 1, "rows e0, e0 + 12 exist for every thread (e0 <= 12), the third is clamped");
            uint32_t mid[NIT]; int lft[NIT], rgt[NIT];
            const int x = x0 - 8 + 4 * sq;
            if (yin) {
                // The frame as a buffer: the thread's column and first row in the vector offset, the tile's row and the 12 rows
                // between a thread's items in the scalar offset -- the twelve loads of a thread share two vector offsets.
                const __amdgpu_buffer_rsrc_t rI = ipx_make_rsrc(frame, (unsigned)(h * w));
                const unsigned srow = (unsigned)((y0 - POLY_N - 1) * w);
                const unsigned dr = x + 4 < w ? 5u : (unsigned)(w - 1 - x);          // right neighbour, from x - 1 (column w is column w - 2)
                const unsigned v01 = (unsigned)(se0 * w + x - 1), v2 = (unsigned)(min(se0 + 2 * QROWS, ER - 1) * w + x - 1);
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const unsigned vo = k < 2 ? v01 : v2, so = k < 2 ? srow + (unsigned)(k * QROWS * w) : srow;
                    lft[k] = (int)__builtin_amdgcn_raw_buffer_load_b8(rI, vo, so, 0);
                    mid[k] = __builtin_amdgcn_raw_buffer_load_b32(rI, vo + 1u, so, 0);
                    rgt[k] = (int)__builtin_amdgcn_raw_buffer_load_b8(rI, vo + dr, so, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int e = min(se0 + QROWS * k, ER - 1);      // (spare lanes repeat the last row)
                    const uint8_t *row = frame + (size_t)reflect101(y0 - POLY_N - 1 + e, h) * w;
                    mid[k] = *reinterpret_cast<const uint32_t *>(row + x);
                    lft[k] = row[x - 1];                            // x0 >= 64 here
                    rgt[k] = row[x + 4 < w ? x + 4 : w - 2];        // reflect-101: column w is column w - 2
                }
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int e = se0 + QROWS * k;
                float pp[6];
                pp[0] = (float)lft[k];
                pp[1] = (float)(mid[k] & 0xff); pp[2] = (float)((mid[k] >> 8) & 0xff);
                pp[3] = (float)((mid[k] >> 16) & 0xff); pp[4] = (float)(mid[k] >> 24);
                pp[5] = (float)rgt[k];
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { float v = pp[c + 1] * k1; v += (pp[c] + pp[c + 2]) * k2; o[c] = v; }
                if (sact && e < ER) *reinterpret_cast<float4 *>(s_rp + e * PX_IP + 4 * sq) = make_float4(o[0], o[1], o[2], o[3]);
            }
        } else {
            for (int i = tid; i < ER * PX_SW; i += 256) {
                const int e = i / PX_SW, c = i - e * PX_SW;
                int x = x0 - POLY_N + c;
                x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
                const uint8_t *row = frame + (size_t)reflect101(y0 - POLY_N - 1 + e, h) * w;
                float v = (float)row[x] * k1;
                v += ((float)row[reflect101(x - 1, w)] + (float)row[reflect101(x + 1, w)]) * k2;
                s_rp[e * PX_IP + PX_OX + c] = v;
            }
        }
        __syncthreads();
        // column pass at the window's rows (replicated beyond the image): window row r <-> image row yc, row-pass rows
        // e = yc - (y0 - 6) and its two neighbours
        if (yin) {
            // window row r = row-pass rows r, r + 1, r + 2: fixed offsets from the thread's first item
            const float *mb = s_rp + (se0 + 1) * PX_IP + 4 * sq;
            float *ob = s_in + se0 * PX_IP + 4 * sq;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (sact && se0 + QROWS * k < PX_SH) {
                    const float *mrow = mb + k * QROWS * PX_IP;
                    const float4 up = *reinterpret_cast<const float4 *>(mrow - PX_IP), md = *reinterpret_cast<const float4 *>(mrow),
                                 dn = *reinterpret_cast<const float4 *>(mrow + PX_IP);
                    float4 o;
                    o.x = k1 * md.x; o.x += k2 * (dn.x + up.x);
                    o.y = k1 * md.y; o.y += k2 * (dn.y + up.y);
                    o.z = k1 * md.z; o.z += k2 * (dn.z + up.z);
                    o.w = k1 * md.w; o.w += k2 * (dn.w + up.w);
                    *reinterpret_cast<float4 *>(ob + k * QROWS * PX_IP) = o;
                }
            }
        } else
        for (int r = se0; sact && r < PX_SH; r += QROWS) {
            const int q = sq;
            int yc = y0 - POLY_N + r;
            yc = yc < 0 ? 0 : (yc > h - 1 ? h - 1 : yc);
            const float *mrow = s_rp + (yc - y0 + POLY_N + 1) * PX_IP + 4 * q;
            const float4 up = *reinterpret_cast<const float4 *>(mrow - PX_IP), md = *reinterpret_cast<const float4 *>(mrow),
                         dn = *reinterpret_cast<const float4 *>(mrow + PX_IP);
            float4 o;
            o.x = k1 * md.x; o.x += k2 * (dn.x + up.x);
            o.y = k1 * md.y; o.y += k2 * (dn.y + up.y);
            o.z = k1 * md.z; o.z += k2 * (dn.z + up.z);
            o.w = k1 * md.w; o.w += k2 * (dn.w + up.w);
            *reinterpret_cast<float4 *>(s_in + r * PX_IP + 4 * q) = o;
        }
    } else if (x0 >= 8 && x0 + PX_IP - 8 <= w && (w & 3) == 0) {
        constexpr int NQ = PX_IP / 4, NIT = (PX_SH * NQ + 255) / 256;
        float4 v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = min(tid + 256 * k, PX_SH * NQ - 1);          // (the spare lanes of the last round repeat the last item)
            const int r = i / NQ, q = i - r * NQ;
            int y = y0 - POLY_N + r;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            v[k] = *reinterpret_cast<const float4 *>(img + (unsigned)(y * w + x0 - 8 + 4 * q));
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + 256 * k;
            const int r = i / NQ, q = i - r * NQ;
            if (i < PX_SH * NQ) *reinterpret_cast<float4 *>(s_in + r * PX_IP + 4 * q) = v[k];
        }
    } else {
        constexpr int NIT = (PX_SH * PX_SW + 255) / 256;
        float v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + 256 * k;
            const int r = i / PX_SW, c = i - r * PX_SW;
            int y = y0 - POLY_N + r, x = x0 - POLY_N + c;
            y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
            x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
            v[k] = i < PX_SH * PX_SW ? img[(unsigned)(y * w + x)] : 0.f;     // a level image has fewer than 2^26 pixels
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + 256 * k;
            const int r = i / PX_SW, c = i - r * PX_SW;
            if (i < PX_SH * PX_SW) s_in[r * PX_IP + PX_OX + c] = v[k];
        }
    }
    __syncthreads();
    // 2. vertical pass.  Columns 0-63 of the window: lane = column, a wave takes four consecutive rows, whose 14 input
    // rows are read once (no index arithmetic, 14 instead of 44 LDS reads); the 10 columns of the right halo
    // (16 x 10 items) go to the first 160 threads.
    auto vpass = [&](const float (&S)[2 * POLY_N + 1], int r, int c) {
        float t0 = S[POLY_N] * pc.g[0], t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int j = 1; j <= POLY_N; ++j) {
            const float s0 = S[POLY_N - j], s1 = S[POLY_N + j];
            const float p = s0 + s1;
            t0 = t0 + pc.g[j] * p;
            t1 = t1 + pc.xg[j] * (s1 - s0);
            t2 = t2 + pc.xxg[j] * p;
        }
        s_t[r][0][c] = t0; s_t[r][1][c] = t1; s_t[r][2][c] = t2;
    };
    {
        static_assert(PX_TH == 16 && PX_TW == 64, "four waves x four consecutive rows x 64 columns");
        const int c = tid & 63, rb = (tid >> 6) * 4;
        float sv[2 * POLY_N + 4];
#pragma unroll
        for (int k = 0; k < 2 * POLY_N + 4; ++k) sv[k] = s_in[(rb + k) * PX_IP + PX_OX + c];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            float S[2 * POLY_N + 1];
#pragma unroll
            for (int k = 0; k <= 2 * POLY_N; ++k) S[k] = sv[rr + k];
            vpass(S, rb + rr, c);
        }
        if (tid < PX_TH * 2 * POLY_N) {
            const int r = tid / (2 * POLY_N), ch = PX_TW + tid - r * (2 * POLY_N);
            float S[2 * POLY_N + 1];
#pragma unroll
            for (int k = 0; k <= 2 * POLY_N; ++k) S[k] = s_in[(r + k) * PX_IP + PX_OX + ch];
            vpass(S, r, ch);
        }
    }
    __syncthreads();
    // 3. horizontal pass: lane = column, wave w takes rows w, w+4, ...
    const int lane = tid & 63, x = x0 + lane;
    if (x >= w) return;
    const size_t P = (size_t)h * w;
    const unsigned P16 = (unsigned)(P * 16);
    const __amdgpu_buffer_rsrc_t rO = ipx_make_rsrc(R + (size_t)f * r_stride(P), (unsigned)(r_stride(P) * 4));
    for (int r = __builtin_amdgcn_readfirstlane(tid >> 6); r < PX_TH; r += 4) {
        const int y = y0 + r;
        if (y >= h) break;
        const float *r0 = s_t[r][0] + lane + POLY_N, *r1 = s_t[r][1] + lane + POLY_N, *r2 = s_t[r][2] + lane + POLY_N;
        double b1 = r0[0] * pc.g[0], b2 = 0, b3 = r1[0] * pc.g[0], b4 = 0, b5 = r2[0] * pc.g[0], b6 = 0;
#pragma unroll
        for (int j = 1; j <= POLY_N; ++j) {
            const double tg = r0[j] + r0[-j];
            const float g0 = pc.g[j];
            b1 = __builtin_fma(tg, pc.gd[j], b1);
            b4 = __builtin_fma(tg, pc.xxgd[j], b4);
            b2 += (r0[j] - r0[-j]) * pc.xg[j];
            b3 += (r1[j] + r1[-j]) * g0;
            b6 += (r1[j] - r1[-j]) * pc.xg[j];
            b5 += (r2[j] + r2[-j]) * g0;
        }
        // expansion of a frame: a float4 plane (coefficients 0-3) followed by a float plane (coefficient 4).  Stored through
        // the frame's buffer: column in the vector offset, row (uniform over the wave) in the scalar offset, no address
        // arithmetic.  Level 0 is 21 GB per batch that nothing reads before every cache has turned over: non-temporal (2).
        const unsigned row = (unsigned)(y * w);
        const ipx_u32x4 ov = {__float_as_uint((float)(b3 * pc.ig11)), __float_as_uint((float)(b2 * pc.ig11)),
                              __float_as_uint((float)(b1 * pc.ig03 + b5 * pc.ig33)), __float_as_uint((float)(b1 * pc.ig03 + b4 * pc.ig33))};
        ipx_buffer_store_b128<FUSED ? 2 : 0>(ov, rO, 16u * (unsigned)x, 16u * row);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint((float)(b6 * pc.ig55)), rO, 4u * (unsigned)x, P16 + 4u * row, FUSED ? 2 : 0);
    }
}

// mean magnitude: the per-workgroup partial sums (f64) of the last iteration, added in a fixed order
__global__ void magnitude_finish_kernel(const double *partial, int nparts, size_t P, int npairs, double thr, float *flows, uint8_t *keep,
                                        const unsigned *err)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += partial[(size_t)p * nparts + i];
    float m = (float)(s / (double)P);                      // np.mean of an f32 array is f32
    if (*err) m = __builtin_nanf("");                      // the stage failed (flow_iter_kernel: consumers_sync): no metric
    flows[p] = m;
    if (keep) keep[p] = (double)m <= thr;                  // ip.py:559 (NaN compares false: discarded)
}

__global__ void first_frame_kernel(float *flows, uint8_t *keep)
{
    flows[0] = 0.f;                                        // ip.py:853: optical_flows[0] = 0
    if (keep) keep[0] = 1;
}

// Overlap probe (debug, IPX_FI_OVERLAP_PROBE=<workgroups>[,<iterations>]): an LDS-free kernel of pure vector
// arithmetic (8 independent f32 chains, 48 registers at most) that run_pairs puts on a side stream beside the
// level-0 iterations, to measure how much arithmetic the memory-bound marching kernel leaves room for on the
// same CUs (its two workgroups hold 158.7 of the CU's 160 KB of LDS: only kernels without LDS can join them).
__global__ __launch_bounds__(256) void fi_probe_burn_kernel(int iters, float *sink)
{
    float a0 = threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 1.0000001f, c = 1e-7f;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a0 = __builtin_fmaf(a0, m, c); a1 = __builtin_fmaf(a1, m, c); a2 = __builtin_fmaf(a2, m, c); a3 = __builtin_fmaf(a3, m, c);
            a4 = __builtin_fmaf(a4, m, c); a5 = __builtin_fmaf(a5, m, c); a6 = __builtin_fmaf(a6, m, c); a7 = __builtin_fmaf(a7, m, c);
        }
    }
    const float s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == 12345.678f) sink[0] = s;                      // never true: keeps the chains alive
}

// the same with f64 chains (add, fma, or convert + add), to measure the issue rate of the f64 instructions the consumers use
template <int KIND>
__global__ __launch_bounds__(256) void fi_probe_burn64_kernel(int iters, float *sink)
{
    double a0 = threadIdx.x, a1 = a0 + 1., a2 = a0 + 2., a3 = a0 + 3., a4 = a0 + 4., a5 = a0 + 5., a6 = a0 + 6., a7 = a0 + 7.;
    const double m = 1.0000001, c = 1e-7;
    float f = (float)threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }
            else if (KIND == 1) {
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
            } else {
                a0 += (double)f; f += 1.f; a1 += (double)f; f += 1.f; a2 += (double)f; f += 1.f; a3 += (double)f; f += 1.f;
            }
        }
    }
    const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (s == 12345.678) sink[0] = (float)s;
}

struct FiProbe { int wgs, iters; hipStream_t side; hipEvent_t fork, join, b0, b1; bool ready; };
FiProbe *fi_probe()
{
    static FiProbe pr = {0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, false};
    static bool parsed = false;
    if (!parsed) {
        parsed = true;
        const char *e = getenv("IPX_FI_OVERLAP_PROBE");
        if (e && *e) {
            pr.wgs = atoi(e);
            const char *c = strchr(e, ',');
            pr.iters = c ? atoi(c + 1) : 20000;
        }
    }
    if (pr.wgs <= 0) return nullptr;
    if (!pr.ready) {
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);          // plo = lowest priority (largest number)
        if (hipStreamCreateWithPriority(&pr.side, hipStreamNonBlocking, getenv("IPX_FI_SIDE_LOW") ? plo : 0) != hipSuccess) return nullptr;
        (void)hipEventCreateWithFlags(&pr.fork, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&pr.join, hipEventDisableTiming);
        (void)hipEventCreate(&pr.b0); (void)hipEventCreate(&pr.b1);
        pr.ready = true;
    }
    return &pr;
}

// ------------------------------------------------------------------------------------------------
int build_plan(ipx_ctx *ctx, int H, int W)
{
    if (ctx->flow_plan && ctx->flow_plan->H == H && ctx->flow_plan->W == W) return IPX_OK;
    // flow_iter_kernel addresses the two expansions of a pair (40 bytes per pixel) with 32-bit buffer offsets
    if ((size_t)H * W > ((size_t)1 << 26)) {
        ipx_set_error("optical flow: frames of more than 2^26 pixels are not supported (got %d x %d)", H, W);
        return IPX_ERR_UNSUPPORTED;
    }
    IPX_CHECK_HIP(hipStreamSynchronize(ctx->stream));       // tables of the old plan may be in use
    ipx_flow_plan_free(ctx->flow_plan);
    ctx->flow_plan = nullptr;
    ipx_flow_plan *pl = new ipx_flow_plan();
    pl->H = H; pl->W = W;
    const double pyr_scale = 0.5;
    int levels = 3;                                         // ip.py:862
    {
        double scale = 1; int k;
        for (k = 0; k < levels; ++k) {
            scale *= pyr_scale;
            if (W * scale < 32 || H * scale < 32) break;
        }
        levels = k;
    }
    pl->levels = levels;
    poly_prepare(POLY_N, 2.0, pl->pc);                      // ip.py:865-866
    std::vector<char> host;
    auto push = [&](const void *src, size_t bytes) -> size_t {
        const size_t off = ipx_align_up(host.size(), 16);
        host.resize(off + bytes);
        memcpy(host.data() + off, src, bytes);
        return off;
    };
    size_t offs[MAX_LEVELS][10];
    for (int k = 0; k <= levels; ++k) {
        LevelGeom &g = pl->lv[k];
        double scale = 1;
        for (int i = 0; i < k; ++i) scale *= pyr_scale;
        const double sigma = (1. / scale - 1) * 0.5;
        int sz = cv_round(sigma * 5) | 1;
        g.ksize = sz > 3 ? sz : 3;
        g.lw = cv_round(W * scale);
        g.lh = cv_round(H * scale);
        if (g.ksize > MAX_KS) { delete pl; ipx_set_error("Gaussian aperture %d unsupported", g.ksize); return IPX_ERR_UNSUPPORTED; }
        gaussian_kernel(g.ksize, sigma, g.kern);
        g.mode = (g.lw == W && g.lh == H) ? 0 : (W == 2 * g.lw && H == 2 * g.lh) ? 1 : 2;
        std::vector<int> xo, yo; std::vector<float> xa, yb;
        linear_table_x(W, g.lw, xo, xa);
        linear_table_y(H, g.lh, yo, yb);
        g.exact_scale = 0;
        for (int sc = 4; sc <= 8 && g.mode == 2 && (W & 3) == 0; sc += 4) {
            bool ok = W == sc * g.lw && H == sc * g.lh;
            for (int d = 0; ok && d < g.lw; ++d) ok = xo[d] == sc * d + sc / 2 - 1 && xa[d] == 0.5f;
            for (int d = 0; ok && d < g.lh; ++d) ok = yo[2 * d] == sc * d + sc / 2 - 1 && yo[2 * d + 1] == sc * d + sc / 2 && yb[d] == 0.5f;
            if (ok) g.exact_scale = sc;
        }
        offs[k][0] = push(xo.data(), xo.size() * 4); offs[k][1] = push(xa.data(), xa.size() * 4);
        offs[k][2] = push(yo.data(), yo.size() * 4); offs[k][3] = push(yb.data(), yb.size() * 4);
        // sample columns / rows of every level pixel, all three resize modes in one form
        std::vector<int> xs2(2 * (size_t)g.lw), ys2(2 * (size_t)g.lh);
        for (int d = 0; d < g.lw; ++d) {
            if (g.mode == 0) xs2[2 * d] = xs2[2 * d + 1] = d;
            else if (g.mode == 1) { xs2[2 * d] = 2 * d; xs2[2 * d + 1] = 2 * d + 1; }
            else { xs2[2 * d] = xo[d]; xs2[2 * d + 1] = xo[d] + 1 > W - 1 ? W - 1 : xo[d] + 1; }
        }
        for (int d = 0; d < g.lh; ++d) {
            if (g.mode == 0) ys2[2 * d] = ys2[2 * d + 1] = d;
            else if (g.mode == 1) { ys2[2 * d] = 2 * d; ys2[2 * d + 1] = 2 * d + 1; }
            else { ys2[2 * d] = yo[2 * d]; ys2[2 * d + 1] = yo[2 * d + 1]; }
        }
        offs[k][8] = push(xs2.data(), xs2.size() * 4); offs[k][9] = push(ys2.data(), ys2.size() * 4);
        // tile: as many level rows as keep the staged source window within 64 rows (one lane per row in
        // the row pass), as many level columns as keep the tile within 26 KB of LDS
        const int Rk = g.ksize / 2;
        auto extent = [&](const std::vector<int> &t2, int n, int tile) {
            int mx = 0;
            for (int d0 = 0; d0 < n; d0 += tile) {
                const int d1 = d0 + tile - 1 < n - 1 ? d0 + tile - 1 : n - 1;
                const int e = t2[2 * d1 + 1] - t2[2 * d0] + 1 + 2 * Rk;
                if (e > mx) mx = e;
            }
            return mx;
        };
        int toy = 1;
        const size_t max_lds = 26 * 1024;                   // six workgroups per CU (measured: 52 KB tiles were 16 % slower)
        while (toy < g.lh && toy < 64 && extent(ys2, g.lh, toy + 1) <= 64) ++toy;
        g.toy = toy;
        g.src_rows = extent(ys2, g.lh, toy);
        const int nsx = g.mode == 0 ? 1 : 2;
        auto lds_for = [&](int tx) {
            const int sp = (extent(xs2, g.lw, tx) + 7) | 1, tp = (nsx * tx) | 1;
            return ((size_t)g.src_rows * (sp + tp) + 2 * tx) * 4 + 16;
        };
        int tox = 64;
        while (tox > 8 && lds_for(tox) > max_lds) tox >>= 1;
        g.tox = tox;
        g.src_cols = extent(xs2, g.lw, tox);
        g.src_pitch = (g.src_cols + 7) | 1;
        g.tmp_pitch = (nsx * tox) | 1;
        g.lds_bytes = (unsigned)lds_for(tox);
        if (g.lds_bytes > 64 * 1024) { delete pl; ipx_set_error("pyramid tile does not fit in LDS (%u bytes)", g.lds_bytes); return IPX_ERR_UNSUPPORTED; }
    }
    for (int k = 0; k < levels; ++k) {                       // flow of level k from level k+1
        std::vector<int> xo, yo; std::vector<float> xa, yb;
        linear_table_x(pl->lv[k + 1].lw, pl->lv[k].lw, xo, xa);
        linear_table_y(pl->lv[k + 1].lh, pl->lv[k].lh, yo, yb);
        offs[k][4] = push(xo.data(), xo.size() * 4); offs[k][5] = push(xa.data(), xa.size() * 4);
        offs[k][6] = push(yo.data(), yo.size() * 4); offs[k][7] = push(yb.data(), yb.size() * 4);
    }
    void *dev;
    int e = ipx_scratch(ctx, SLOT_FLOW_TAB, host.size(), &dev);
    if (e) { delete pl; return e; }
    hipError_t he = hipMemcpyAsync(dev, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);   // `host` dies at return
    if (he != hipSuccess) { delete pl; ipx_set_error("table upload failed: %s", hipGetErrorString(he)); return IPX_ERR_HIP; }
    const char *d = (const char *)dev;
    for (int k = 0; k <= levels; ++k) {
        LevelGeom &g = pl->lv[k];
        g.x_ofs = (const int *)(d + offs[k][0]); g.x_a = (const float *)(d + offs[k][1]);
        g.y_ofs = (const int *)(d + offs[k][2]); g.y_b = (const float *)(d + offs[k][3]);
        g.xs2 = (const int *)(d + offs[k][8]); g.ys2 = (const int *)(d + offs[k][9]);
        if (k < levels) {
            g.ux_ofs = (const int *)(d + offs[k][4]); g.ux_a = (const float *)(d + offs[k][5]);
            g.uy_ofs = (const int *)(d + offs[k][6]); g.uy_b = (const float *)(d + offs[k][7]);
        }
    }
        const hipError_t ae = ipx_flow_iter_prepare();
    if (ae != hipSuccess) { delete pl; ipx_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(ae)); return IPX_ERR_HIP; }
    ctx->flow_plan = pl;
    return IPX_OK;
}

// Level image + polynomial expansion of launch frames [0, nf) (chunk frames first..first+nf-1).
int launch_level_expansion(ipx_ctx *ctx, const LevelGeom &g, const uint8_t *frames, const uint8_t *extra, int first,
                           int nf, float *I, float *R)
{
    const ipx_flow_plan *pl = ctx->flow_plan;
    PyrArgs a;
    a.frames = frames; a.extra = extra; a.first = first; a.H = pl->H; a.W = pl->W; a.lh = g.lh; a.lw = g.lw;
    a.mode = g.mode; a.tox = g.tox; a.toy = g.toy; a.src_pitch = g.src_pitch; a.tmp_pitch = g.tmp_pitch;
    a.src_rows = g.src_rows; a.xs2 = g.xs2; a.x_a = g.x_a; a.ys2 = g.ys2; a.y_b = g.y_b; a.out = I;
    memcpy(a.kt.k, g.kern, sizeof(a.kt.k));
    a.tiles_x = (g.lw + g.tox - 1) / g.tox; a.tiles_y = (g.lh + g.toy - 1) / g.toy; a.nf = nf;
    const dim3 grid(8 * ((a.tiles_x * a.tiles_y * nf + 7) / 8)), block(256);
    const double pyr_bytes = (double)nf * ((double)pl->H * pl->W + 4.0 * g.lh * g.lw);
    // level 0 of the usual pyramid is not made at all: polyexp_kernel<true> evaluates it from the frames
    const bool fused = g.ksize == 3 && g.mode == 0 && (pl->W & 3) == 0 && (pl->H & 3) == 0;
    if (fused) {
        PxSrc src = {frames, extra, first, g.kern[1], g.kern[2]};
        const int ptx = (g.lw + PX_TW - 1) / PX_TW, pty = (g.lh + PX_TH - 1) / PX_TH;
        IPX_LAUNCH(ctx, KID_POLYEXP, 21.0 * nf * g.lh * g.lw, polyexp_kernel<true>, dim3(8 * ((ptx * pty * nf + 7) / 8)), block, (const float *)nullptr,
                   g.lh, g.lw, ptx, pty, nf, pl->pc, R, src);
        IPX_LAUNCH_CHECK();
        return IPX_OK;
    }
    if (ctx->prof.on) ipx_prof_begin(ctx, KID_PYR, pyr_bytes);
    if (g.ksize == 3 && g.mode <= 1 && (pl->W & 3) == 0 && (pl->H & 3) == 0) {
        const dim3 g3((pl->W / 4 + 63) / 64, (pl->H / 4 + 3) / 4, nf);
        if (g.mode == 0) hipLaunchKernelGGL(pyr3_kernel<0>, g3, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL(pyr3_kernel<1>, g3, block, 0, ctx->stream, a);
    } else if (g.exact_scale == 4 && g.ksize == 9) {
        a.tiles_x = (g.lw + 31) / 32; a.tiles_y = (g.lh + 7) / 8;
        hipLaunchKernelGGL((pyr_exact_kernel<9, 4, 32, 8>), dim3(8 * ((a.tiles_x * a.tiles_y * nf + 7) / 8)), block, 0, ctx->stream, a);
    } else if (g.exact_scale == 8 && g.ksize == 19) {
        a.tiles_x = (g.lw + 7) / 8; a.tiles_y = (g.lh + 7) / 8;
        hipLaunchKernelGGL((pyr_exact_kernel<19, 8, 8, 8>), dim3(8 * ((a.tiles_x * a.tiles_y * nf + 7) / 8)), block, 0, ctx->stream, a);
    } else
    switch (g.ksize) {
    case 3: hipLaunchKernelGGL(pyr_level_kernel<3>, grid, block, g.lds_bytes, ctx->stream, a); break;
    case 9: hipLaunchKernelGGL(pyr_level_kernel<9>, grid, block, g.lds_bytes, ctx->stream, a); break;
    case 19: hipLaunchKernelGGL(pyr_level_kernel<19>, grid, block, g.lds_bytes, ctx->stream, a); break;
    default:
        ipx_set_error("Gaussian aperture %d unsupported", g.ksize);
        return IPX_ERR_UNSUPPORTED;
    }
    if (ctx->prof.on) ipx_prof_end(ctx);
    const int ptx = (g.lw + PX_TW - 1) / PX_TW, pty = (g.lh + PX_TH - 1) / PX_TH;
    IPX_LAUNCH(ctx, KID_POLYEXP, 24.0 * nf * g.lh * g.lw, polyexp_kernel<false>, dim3(8 * ((ptx * pty * nf + 7) / 8)), block, I, g.lh, g.lw, ptx, pty, nf, pl->pc, R,
               PxSrc{nullptr, nullptr, 0, 0.f, 0.f});
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

// Rows of a level are cut into segments so that a launch has a few thousand workgroups.
struct FiGeom { int strips, nseg, seg_rows, ppx; };
FiGeom fi_geometry(const LevelGeom &g, int np)
{
    FiGeom f;
    f.strips = (g.lw + FI_TW - 1) / FI_TW;
    // One segment per strip: only a march that starts at row 0 reproduces the reference's running vertical
    // sums bit for bit (flow_iter_kernel).  The kernel itself can cut a strip into several segments; that was
    // measured to change nothing for the 256-pair batches of a chunk (1024 to 8192 workgroups per launch).
    (void)np;
    f.nseg = 1;
    f.seg_rows = (g.lh + FI_G - 1) / FI_G * FI_G;
    f.ppx = (np + 7) / 8;                                   // pairs per XCD
    return f;
}

// partial magnitude sums per pair of run_pairs(..., uv_out = nullptr, partial)
int fi_partials(const ipx_flow_plan *pl, int np)
{
    const FiGeom f = fi_geometry(pl->lv[0], np);
    return f.strips * f.nseg;
}

// scratch of a batch of np pairs: level images + expansions of np + 1 frames; two flow fields per pair
size_t flow_pyr_bytes(int np, size_t P0) { return ((size_t)(np + 1) * P0 + 64 + (size_t)(np + 1) * r_stride(P0)) * sizeof(float) + 512; }
size_t flow_uv_bytes(int np, size_t P0) { return 2 * (size_t)np * P0 * sizeof(float2) + 512; }

// Flow of `np` pairs; pair p = launch frames (p, p+1).  Either the level-0 field (*uv_out) or, when
// uv_out is null, `fi_partials` partial sums of its magnitudes per pair in `partial`.
int run_pairs(ipx_ctx *ctx, const uint8_t *frames, const uint8_t *extra, int first, int np, float2 **uv_out, double *partial,
              unsigned *err)
{
    const ipx_flow_plan *pl = ctx->flow_plan;
    const int nf = np + 1;
    const size_t P0 = (size_t)pl->lv[0].lh * pl->lv[0].lw;
    void *pyr, *uv;
    IPX_TRY(ipx_scratch(ctx, SLOT_FLOW_PYR, flow_pyr_bytes(np, P0), &pyr));
    IPX_TRY(ipx_scratch(ctx, SLOT_FLOW_UV, flow_uv_bytes(np, P0), &uv));
    // two flow buffers, every iteration reads `cur` and writes `other`; the first iteration of a level
    // reads the coarser level's result (or nothing at the coarsest level)
    float2 *cur = (float2 *)uv, *other = cur + ipx_align_up((size_t)np * P0, 32);
    for (int k = pl->levels; k >= 0; --k) {
        const LevelGeom &g = pl->lv[k];
        const size_t P = (size_t)g.lh * g.lw;
        float *I = (float *)pyr, *R = I + ipx_align_up((size_t)nf * P, 64);
        IPX_TRY(launch_level_expansion(ctx, g, frames, extra, first, nf, I, R));
        const FiGeom f = fi_geometry(g, np);
        const dim3 fgrid(8 * f.ppx * f.strips * f.nseg);
        FiArgs a;
        a.R = R; a.partial = partial;
        a.h = g.lh; a.w = g.lw; a.seg_rows = f.seg_rows; a.strips = f.strips; a.nseg = f.nseg; a.np = np; a.ppx = f.ppx;
        a.sh = a.sw = 0; a.x_ofs = a.y_ofs = nullptr; a.x_a = a.y_b = nullptr;
        a.err = err; a.spin_limit = ctx->fi_spin_limit;
        // Workgroup order of an XCD: three neighbouring strips of a third of its pairs at a time (flow_iter_kernel).  Against one
        // strip of all its pairs, 512 pairs of 1080p: 36.8 instead of 38.5 GB fetched by a level-0 launch (29.7 / 33.4 GB when the
        // coarser field is read), the 12 launches of a batch 41.2 instead of 42.3 ms; 2, 4 and 6 strips measured the same to 0.3 %.
        static const int env_sgroup = getenv("IPX_FI_STRIP_GROUP") ? atoi(getenv("IPX_FI_STRIP_GROUP")) : 3;   // (A/B runs: tools/ab_sgroup.sh)
        a.sgroup = env_sgroup < 1 ? 1 : env_sgroup;
        a.prio = getenv("IPX_FI_PRIO") ? atoi(getenv("IPX_FI_PRIO")) : 0;   // probe: 1 = all waves, 2 = consumers, 3 = producers at high priority
        FiProbe *probe = k == 0 ? fi_probe() : nullptr;
        const bool probe_alone = probe && getenv("IPX_FI_OVERLAP_ALONE") != nullptr;   // the burn without the iterations beside it
        if (probe) {
            (void)hipEventRecord(probe->fork, ctx->stream);
            (void)hipStreamWaitEvent(probe->side, probe->fork, 0);
            (void)hipEventRecord(probe->b0, probe->side);
            const char *k64 = getenv("IPX_FI_OVERLAP_F64");
            if (!k64) hipLaunchKernelGGL(fi_probe_burn_kernel, dim3(probe->wgs), dim3(256), 0, probe->side, probe->iters, (float *)uv);
            else if (k64[0] == '0') hipLaunchKernelGGL(fi_probe_burn64_kernel<0>, dim3(probe->wgs), dim3(256), 0, probe->side, probe->iters, (float *)uv);
            else if (k64[0] == '1') hipLaunchKernelGGL(fi_probe_burn64_kernel<1>, dim3(probe->wgs), dim3(256), 0, probe->side, probe->iters, (float *)uv);
            else hipLaunchKernelGGL(fi_probe_burn64_kernel<2>, dim3(probe->wgs), dim3(256), 0, probe->side, probe->iters, (float *)uv);
            (void)hipEventRecord(probe->b1, probe->side);
            (void)hipEventRecord(probe->join, probe->side);
            if (probe_alone) (void)hipStreamWaitEvent(ctx->stream, probe->join, 0);
        }
        auto fi_launch = [&](int in_mode, bool red, double abytes) {  // (the kernel lives in flow_iter.hip)
            if (ctx->prof.on) ipx_prof_begin(ctx, KID_FLOW_ITER, abytes, k);
            ipx_flow_iter_launch(ctx->stream, in_mode, red, fgrid, a);
            if (ctx->prof.on) ipx_prof_end(ctx);
        };
        for (int it = 0; it < 3; ++it) {                            // ip.py:864 iterations=3
            a.flow_in = cur; a.flow_out = other;
            const bool reduce = !uv_out && k == 0 && it == 2;
            // algorithmic bytes of the launch: every array once.  The np pairs of a batch share their
            // expansions (frame p+1 is the second frame of pair p and the first of pair p+1): np + 1 expansions of
            // 20 bytes per pixel, not 2 np; per pair the flow written (8) and the flow read (8, or the coarser field).
            double bytes = 20.0 * P * (np + 1) + (reduce ? 0.0 : 8.0 * P * np);
            if (it > 0) {
                bytes += 8.0 * P * np;
                fi_launch(0, reduce, bytes);
            } else if (k == pl->levels) {
                fi_launch(1, false, bytes);
            } else {
                const LevelGeom &c = pl->lv[k + 1];
                a.sh = c.lh; a.sw = c.lw; a.x_ofs = g.ux_ofs; a.x_a = g.ux_a; a.y_ofs = g.uy_ofs; a.y_b = g.uy_b;
                bytes += 8.0 * c.lh * c.lw * np;
                fi_launch(2, false, bytes);
            }
            float2 *t = cur; cur = other; other = t;
        }
        if (probe) {
            if (!probe_alone) (void)hipStreamWaitEvent(ctx->stream, probe->join, 0);
            (void)hipEventSynchronize(probe->join);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, probe->b0, probe->b1);
            fprintf(stderr, "[fi probe] burn %d workgroups x %d iterations: %.3f ms (%s the level-0 iterations)\n", probe->wgs, probe->iters, ms,
                    probe_alone ? "before" : "beside");
        }
        IPX_LAUNCH_CHECK();
    }
    if (uv_out) *uv_out = cur;
    return IPX_OK;
}

// pairs per batch: bounded by a scratch budget
int pair_batch(int H, int W, int want)
{
    const size_t per_pair = (size_t)H * W * (6 + 4) * sizeof(float);
    static const long env_pairs = getenv("IPX_FLOW_BATCH_PAIRS") ? atol(getenv("IPX_FLOW_BATCH_PAIRS")) : 0;   // (A/B runs)
    const size_t cap_pairs = env_pairs > 0 ? (size_t)env_pairs : 512, cap_bytes = env_pairs > 0 ? ((size_t)200 << 30) : ((size_t)48 << 30);
    size_t b = cap_bytes / per_pair;                     // 288 GB of HBM: 48 GB of scratch buys 512-pair batches at 1080p, 144 at 4K
    if (b < 1) b = 1;
    if (b > cap_pairs) b = cap_pairs;                    // measured at 1080p: 64 -> 256 pairs per batch +3 %, 256 -> 512 +1.6 %, 1024 +2.0 %
    if (b > (size_t)want) b = want;
    return (int)b;
}

// SLOT_FLOW_RED = [error word (256 bytes) | partial sums].  The word starts at zero and stays raised once a kernel
// has raised it: sticky until ipx_flow_status(..., clear), also when the slot has to grow (the word of the old arena
// is read before that arena is freed and carried over into the new one).
int flow_red(ipx_ctx *ctx, size_t partial_bytes, void **partial, unsigned **err)
{
    const bool fresh = ctx->slot_bytes[SLOT_FLOW_RED] < partial_bytes + 256;
    unsigned carried = 0;
    if (fresh && ctx->flow_ran && ctx->slot[SLOT_FLOW_RED]) {
        IPX_CHECK_HIP(hipMemcpyAsync(&carried, ctx->slot[SLOT_FLOW_RED], sizeof(carried), hipMemcpyDeviceToHost, ctx->stream));
        IPX_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    void *base;
    IPX_TRY(ipx_scratch(ctx, SLOT_FLOW_RED, partial_bytes + 256, &base));
    if (fresh) {
        IPX_CHECK_HIP(hipMemsetAsync(base, 0, 256, ctx->stream));
        if (carried) IPX_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)base, (int)carried, 1, ctx->stream));
    }
    *err = (unsigned *)base;
    *partial = (char *)base + 256;
    ctx->flow_ran = 1;
    return IPX_OK;
}

}  // namespace

// Reads the flow stage's error word (synchronises the stream).  *raised = 1 when a flow_iter hand-off has timed out
// since the word was last cleared.
extern "C" int ipx_flow_status(ipx_ctx *ctx, int *raised, int clear)
{
    IPX_REQUIRE(ctx && raised, "null pointer");
    IPX_TRY(ipx_ctx_enter(ctx));
    *raised = 0;
    if (!ctx->flow_ran || !ctx->slot[SLOT_FLOW_RED]) return IPX_OK;
    unsigned w = 0;
    IPX_CHECK_HIP(hipMemcpyAsync(&w, ctx->slot[SLOT_FLOW_RED], sizeof(w), hipMemcpyDeviceToHost, ctx->stream));
    IPX_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *raised = w != 0;
    if (w && clear) IPX_CHECK_HIP(hipMemsetAsync(ctx->slot[SLOT_FLOW_RED], 0, sizeof(w), ctx->stream));
    return IPX_OK;
}

extern "C" int ipx_debug_fi_spin_limit(ipx_ctx *ctx, unsigned limit)
{
    IPX_REQUIRE(ctx, "null pointer");
    ctx->fi_spin_limit = limit;
    return IPX_OK;
}

// The plan (its tables are uploaded) and the two large arenas of the stage for chunks of up to n frames of H x W, made
// now rather than by the first ipx_flow_metric call.
extern "C" int ipx_reserve_flow(ipx_ctx *ctx, int n, int H, int W)
{
    IPX_REQUIRE(ctx, "null pointer");
    IPX_REQUIRE(n > 0 && H >= 2 && W >= 2, "bad sizes (frames must be at least 2x2)");
    IPX_TRY(ipx_ctx_enter(ctx));
    IPX_TRY(build_plan(ctx, H, W));
    const int B = pair_batch(H, W, n > 1 ? n - 1 : 1);
    const LevelGeom &g0 = ctx->flow_plan->lv[0];
    const size_t P0 = (size_t)g0.lh * g0.lw;
    void *p;
    IPX_TRY(ipx_scratch(ctx, SLOT_FLOW_PYR, flow_pyr_bytes(B, P0), &p));
    IPX_TRY(ipx_scratch(ctx, SLOT_FLOW_UV, flow_uv_bytes(B, P0), &p));
    return IPX_OK;
}

extern "C" int ipx_flow_metric(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W,
                               const uint8_t *prev_frame, float *flows, uint8_t *keep)
{
    IPX_REQUIRE(ctx && frames && flows, "null pointer");
    IPX_REQUIRE(n > 0 && H >= 2 && W >= 2, "bad sizes (frames must be at least 2x2)");
    IPX_TRY(ipx_ctx_enter(ctx));
    IPX_TRY(build_plan(ctx, H, W));
    const size_t P = (size_t)H * W;
    const int first_pair = prev_frame ? 0 : 1;       // pair i = (frame i-1, frame i)
    if (!prev_frame) IPX_LAUNCH(ctx, KID_MISC, 5, first_frame_kernel, dim3(1), dim3(1), flows, keep);
    const int total = n - first_pair;
    if (total <= 0) { IPX_LAUNCH_CHECK(); return IPX_OK; }
    const int B = pair_batch(H, W, total);
    void *red;
    const LevelGeom &g0 = ctx->flow_plan->lv[0];             // upper bound of fi_partials for any batch size
    const size_t max_parts = (size_t)((g0.lw + FI_TW - 1) / FI_TW) * ((g0.lh + 63) / 64);
    unsigned *err;
    IPX_TRY(flow_red(ctx, (size_t)B * max_parts * sizeof(double), &red, &err));
    for (int i0 = first_pair; i0 < n; i0 += B) {
        const int np = n - i0 < B ? n - i0 : B;
        const int nparts = fi_partials(ctx->flow_plan, np);
        IPX_TRY(run_pairs(ctx, frames, prev_frame, i0 - 1, np, nullptr, (double *)red, err));
        IPX_LAUNCH(ctx, KID_MISC, 8.0 * nparts * np, magnitude_finish_kernel, dim3((np + 63) / 64), dim3(64), (const double *)red, nparts, P, np,
                           ctx->p.flow_threshold, flows + i0, keep ? keep + i0 : nullptr, (const unsigned *)err);
        IPX_LAUNCH_CHECK();
    }
    return IPX_OK;
}


namespace {
__global__ void f32_to_u8_kernel(const float *src, size_t n, uint8_t *dst, int *bad)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = src[i];
    const int q = (int)v;
    if (!(v >= 0.f && v <= 255.f) || (float)q != v) atomicOr(bad, 1);
    dst[i] = (uint8_t)q;
}
}  // namespace

// Unit-parity entry point: the reference hands cv2 float32 copies of uint8 frames (ip.py:856-857),
// so the images must hold integers in [0,255]; anything else is rejected rather than rounded.
extern "C" int ipx_farneback_pair(ipx_ctx *ctx, const float *prev, const float *next, int H, int W, float *flow)
{
    IPX_REQUIRE(ctx && prev && next && flow, "null pointer");
    IPX_REQUIRE(H >= 2 && W >= 2, "bad sizes (frames must be at least 2x2)");
    IPX_TRY(ipx_ctx_enter(ctx));
    IPX_TRY(build_plan(ctx, H, W));
    const size_t P = (size_t)H * W;
    void *st;
    IPX_TRY(ipx_scratch(ctx, SLOT_STAGE, 2 * P + 256, &st));
    uint8_t *u8 = (uint8_t *)st;
    int *bad = (int *)(u8 + ipx_align_up(2 * P, 16));
    IPX_CHECK_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
    const dim3 grid((unsigned)((P + 255) / 256)), block(256);
    IPX_LAUNCH(ctx, KID_MISC, 5.0 * P, f32_to_u8_kernel, grid, block, prev, P, u8, bad);
    IPX_LAUNCH(ctx, KID_MISC, 5.0 * P, f32_to_u8_kernel, grid, block, next, P, u8 + P, bad);
    IPX_LAUNCH_CHECK();
    int hbad = 0;
    IPX_CHECK_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    IPX_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (hbad) {
        ipx_set_error("ipx_farneback_pair: images must hold integer values in [0,255] (float copies of uint8 frames)");
        return IPX_ERR_UNSUPPORTED;
    }
    float2 *uv;
    void *red;
    unsigned *err;
    IPX_TRY(flow_red(ctx, 256, &red, &err));
    IPX_TRY(run_pairs(ctx, u8, nullptr, 0, 1, &uv, nullptr, err));
    IPX_CHECK_HIP(hipMemcpyAsync(flow, uv, P * sizeof(float2), hipMemcpyDeviceToDevice, ctx->stream));
    return IPX_OK;
}


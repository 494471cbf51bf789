// flow_iter.hip -- the marching kernel of the Farneback iterations (see flow.hip for the stage around it).  A translation
// unit of its own because it is built with -mllvm -amdgpu-sched-strategy=max-ilp (image_processing_amd/build.py):
// that scheduler takes 1.5 % off this kernel and adds 7 % to polyexp_kernel, which stays with the default in flow.hip.
#include "ipx_internal.h"
#include "flow_iter.h"
#include <math.h>
#include <type_traits>
#include <utility>

namespace {

// FarnebackUpdateMatrices at one pixel: the five entries (G11, G12, G22, h1, h2) from the
// expansions R0 (at the pixel) and R1 (bilinear sample at pixel + flow), with border damping.
// Split into the loads and the arithmetic so that a caller can keep the loads in flight.
struct UmIn {
    float r0[5];        // R0 at the pixel
    float q[5][4];      // R1 at (x1,y1), (x1+1,y1), (x1,y1+1), (x1+1,y1+1); unset when the sample is out of bounds
    float fx, fy;       // fractions of the sample position (um_position), kept from the loads to the arithmetic
    bool inb;           // the sample is inside the second image
};

// bilinear sample position of pixel (x,y) displaced by d: integer corner, fractions, in-bounds test
__device__ __forceinline__ bool um_position(float2 d, int x, int y, int h, int w, int &x1, int &y1, float &fx, float &fy)
{
    fx = x + d.x; fy = y + d.y;
    x1 = (int)floorf(fx); y1 = (int)floorf(fy);
    fx -= x1; fy -= y1;
    return (unsigned)x1 < (unsigned)(w - 1) && (unsigned)y1 < (unsigned)(h - 1);
}

// Buffer addressing (SGPR resource + SGPR plane offset + one 32-bit VGPR offset per position) keeps the
// address arithmetic of the 16 loads of a matrix pixel out of the vector ALU and out of the VGPR file.

// rR: the expansions of the pair's two frames (layout: polyexp_kernel), FS = bytes per frame, P16 = byte
// offset of the fifth coefficient's plane.  16-byte loads fetch coefficients 0-3 of one pixel; the loads are
// unconditional (no divergent control flow between issue and use).  A position that must not be read (a matrix
// column or row that feeds nothing, a sample outside the image) is replaced by pixel index UM_DEAD ONCE, before
// the offsets are derived from it.  UM_DEAD = 0x3ffffff0 puts BOTH derived offsets beyond any resource this kernel
// is given (frames have at most 2^26 pixels, a resource at most 40 * 2^26 = 0xa0000000 bytes): the 4-byte plane's
// offset is 4 * UM_DEAD = 0xffffffc0 and the 16-byte planes' is 16 * UM_DEAD mod 2^32 = 0xffffff00, so the range check
// (which sees the vector offset alone, not the scalar one) fails for the loads at the position itself and they fetch
// nothing.  The loads one ROW further down add w * 16 resp. w * 4 in 32-bit arithmetic and wrap to a small offset:
// they read a few bytes near the start of the second frame's planes (inside the resource and the allocation:
// harmless) and what they return is never used.  The +16 / +1 row steps come after the select, so they fold into the
// instructions' immediate offsets or cost one addition for both planes.
// (Round 2 used 0x0f000000, whose 4-byte offset 0x3c000000 is INSIDE the resource of frames above 25.2 M pixels and
// was then displaced by the scalar offset to beyond the allocation: ADVICE r2, tests/test_gpu_chunk.py::test_flow_above_25_megapixels.)
constexpr unsigned UM_DEAD = 0x3ffffff0u;
__device__ __forceinline__ void um_load(__amdgpu_buffer_rsrc_t rR, unsigned FS, unsigned P16, float2 d, int x, int y, int h, int w,
                                        UmIn &in, bool live = true)
{
    const unsigned o1 = live ? (unsigned)(y * w + x) : UM_DEAD;
    int x1, y1;
#if IPX_FI_ABLATE & 2
    FS = 0;
#endif
    in.inb = um_position(d, x, y, h, w, x1, y1, in.fx, in.fy);
    const ipx_u32x4 c03 = __builtin_amdgcn_raw_buffer_load_b128(rR, o1 * 16u, 0, 0);
    in.r0[4] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rR, o1 * 4u, P16, 0));
    in.r0[0] = __uint_as_float(c03.x); in.r0[1] = __uint_as_float(c03.y);
    in.r0[2] = __uint_as_float(c03.z); in.r0[3] = __uint_as_float(c03.w);
    const unsigned q1 = in.inb && live ? (unsigned)(y1 * w + x1) : UM_DEAD;
    const unsigned qa = q1 * 16u, qb = qa + 16u * (unsigned)w, qe = q1 * 4u;
    const ipx_u32x4 s00 = __builtin_amdgcn_raw_buffer_load_b128(rR, qa, FS, 0);
#if IPX_FI_ABLATE & 8
    const ipx_u32x4 s01 = s00, s10 = s00, s11 = s00;
    (void)qb;
    const ipx_u32x2 e0 = __builtin_amdgcn_raw_buffer_load_b64(rR, qe, FS + P16, 0);
    const ipx_u32x2 e1 = e0;
#else
    const ipx_u32x4 s01 = __builtin_amdgcn_raw_buffer_load_b128(rR, qa + 16u, FS, 0);
    const ipx_u32x4 s10 = __builtin_amdgcn_raw_buffer_load_b128(rR, qb, FS, 0);
    const ipx_u32x4 s11 = __builtin_amdgcn_raw_buffer_load_b128(rR, qb + 16u, FS, 0);
    const ipx_u32x2 e0 = __builtin_amdgcn_raw_buffer_load_b64(rR, qe, FS + P16, 0);
    const ipx_u32x2 e1 = __builtin_amdgcn_raw_buffer_load_b64(rR, qe + 4u * (unsigned)w, FS + P16, 0);
#endif
    in.q[0][0] = __uint_as_float(s00.x); in.q[0][1] = __uint_as_float(s01.x); in.q[0][2] = __uint_as_float(s10.x); in.q[0][3] = __uint_as_float(s11.x);
    in.q[1][0] = __uint_as_float(s00.y); in.q[1][1] = __uint_as_float(s01.y); in.q[1][2] = __uint_as_float(s10.y); in.q[1][3] = __uint_as_float(s11.y);
    in.q[2][0] = __uint_as_float(s00.z); in.q[2][1] = __uint_as_float(s01.z); in.q[2][2] = __uint_as_float(s10.z); in.q[2][3] = __uint_as_float(s11.z);
    in.q[3][0] = __uint_as_float(s00.w); in.q[3][1] = __uint_as_float(s01.w); in.q[3][2] = __uint_as_float(s10.w); in.q[3][3] = __uint_as_float(s11.w);
    in.q[4][0] = __uint_as_float(e0.x); in.q[4][1] = __uint_as_float(e0.y); in.q[4][2] = __uint_as_float(e1.x); in.q[4][3] = __uint_as_float(e1.y);
}

__device__ __forceinline__ void um_finish(const UmIn &in, float2 d, int x, int y, int h, int w, float (&m)[5])
{
#if IPX_FI_ABLATE & 4
    for (int c = 0; c < 5; ++c) m[c] = (in.r0[c] + in.q[c][0]) + (in.q[c][1] + in.q[c][2]) + (in.q[c][3] + d.x);
    return;
#endif
    const float dx = d.x, dy = d.y;
    const float fx = in.fx, fy = in.fy;
    const bool inb = in.inb;
    float r2, r3, r4, r5, r6;
    if (inb) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        r2 = a00 * in.q[0][0] + a01 * in.q[0][1] + a10 * in.q[0][2] + a11 * in.q[0][3];
        r3 = a00 * in.q[1][0] + a01 * in.q[1][1] + a10 * in.q[1][2] + a11 * in.q[1][3];
        r4 = a00 * in.q[2][0] + a01 * in.q[2][1] + a10 * in.q[2][2] + a11 * in.q[2][3];
        r5 = a00 * in.q[3][0] + a01 * in.q[3][1] + a10 * in.q[3][2] + a11 * in.q[3][3];
        r6 = a00 * in.q[4][0] + a01 * in.q[4][1] + a10 * in.q[4][2] + a11 * in.q[4][3];
        r4 = (in.r0[2] + r4) * 0.5f;
        r5 = (in.r0[3] + r5) * 0.5f;
        r6 = (in.r0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = in.r0[2];
        r5 = in.r0[3];
        r6 = in.r0[4] * 0.5f;
    }
    r2 = (in.r0[0] - r2) * 0.5f;
    r3 = (in.r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    constexpr int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(w - BORDER * 2) || (unsigned)(y - BORDER) >= (unsigned)(h - BORDER * 2)) {
        auto bw = [](int i) -> float { return i < 2 ? 0.14f : 0.4472f; };   // {0.14, 0.14, 0.4472, 0.4472, 0.4472}
        const float scale = (x < BORDER ? bw(x) : 1.f) * (x >= w - BORDER ? bw(w - x - 1) : 1.f) *
                            (y < BORDER ? bw(y) : 1.f) * (y >= h - BORDER ? bw(h - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// One Farneback iteration, fused: FarnebackUpdateMatrices -> 15x15 box window (replicate borders,
// f64) -> regularised 2x2 solve.  The matrices never touch HBM.  A 512-thread workgroup owns a strip of
// 114 output columns (128 matrix columns with the window halo) and MARCHES down a segment of rows, 4 rows
// per step, through a 23-row LDS ring of matrix rows.  The waves are specialised so that the memory phase
// of one step runs under the arithmetic of the previous one:
//   producers (waves 4-7): two matrix rows per thread, row-pipelined: while one row is finished and stored
//      into the ring, the loads of the other and of the row that replaces the finished one are in flight;
//   B  (waves 0-3) threads own (column, channel group) items and keep the running vertical 15-row sums in f64
//      registers (+ new row - row leaving the window), writing the 4 rows of sums to LDS;
//   C  (waves 0-3, one row each) horizontal 15-sums slide over two adjacent outputs per thread, then the
//      2x2 solve and the store (or the magnitude sum).
// One workgroup barrier per step (plus an LDS counter on which the four consumer waves meet between B and C);
// the ring is large enough (rows yb-8 .. yb+14) that the producers never write a row B still reads.
// 79,376 bytes of LDS: two workgroups per CU.  Only the horizontal halo (128/114) and 18 rows per segment
// are computed twice.
// Measured on MI355X (in-kernel stamps, PMC, ablation builds: tools/fi_stamps.py, tools/pmc_stage.sh,
// tools/fi_ablate.py; 1080p, two workgroups per CU).  A step takes ~4.3k cycles (round 2 began at 5.5k).  No single
// resource sets it: the vector ALUs are 77 % busy (~3.0 wave instructions per output pixel: producers 52 %, C 30 %,
// B 18 %), the LDS 52 %, and three floors lie close together -- the ALU work alone (77 % of the time), the
// producers alone (consumer waves idle: 75 %; they are then paced by the 108 bytes per matrix pixel that the
// 16 loads pull through L1: loading one bilinear tap instead of four takes another 18 % off, replacing their
// arithmetic only 4 %) and HBM (one tap, no consumers: 6.2 TB/s, the streaming rate of the device; 60 % of the
// time).  Sampling the second expansion from the pair's first frame (its lines L1 hits) changes 5 %: the misses
// are not what the kernel waits for.  The two wave roles are balanced to within 8 % (producers 3.7k busy cycles per
// step, consumers 3.4k); the rest of a step is the skew of eight waves meeting at the barrier.
// What moved the time in round 2 (5.5k -> 4.3k): phase C was the critical path -- its ds_bpermute exchanges (4 LDS
// cycles per 32 bits moved) are now 8-byte reads and writes in place (see there), the reads spelled as ds_read_b64
// because the compiler's ds_read2_b64 pairs take the LDS array twice as long; phase B without branches, all ring
// reads of a step in flight at once; the producers with one dead-offset select per position instead of one per
// load, the sample fractions kept from the loads to the arithmetic, running ring slots (-14 % of their vector
// instructions).  Earlier: fewer L1 line requests per pixel (float4 expansion planes: 8 loads instead of 15),
// keeping requests queued at all times (row pipeline), taking B and the B->C hand-off off the producers, and
// nothing in the producers' loop that waits for a load it does not need: the loop must not begin with loads in
// flight that the compiler ordered differently from the loop's own (its wait counts are the worst case over both
// entries), no loaded value may be copied at the loop's end, and tables that are uniform per wave must be scalar
// loads (see row_tab).
// What did not: LDS counters in place of the step barrier (producers wait for B of the step before, consumers for
// the producers' stores and for C of the step before, every counter read issued ahead of its use: +3 %, the
// polls cost what the skew did); a second workgroup barrier in place of the B -> C counter (+2 %: the producers'
// loads leave later); the in-bounds test of the sample as selects instead of a branch (+1.5 %); wave priorities
// (consumers or producers raised: +-0); padding the rows of vertical sums so that B's 8-byte stores hit distinct
// banks (the two-way conflicts are 15 % of the LDS cycles and cost nothing); deeper prefetch, 64-column strips,
// one output per thread in C, and 1024-thread workgroups whose two halves take consecutive pairs in lockstep so
// that the shared expansion is fetched once per CU (the coupled barrier costs more than the saved misses return).
// Later in round 2 (final build: ~4.05k cycles per step, vector ALUs 81 % busy at 1.91 GHz, DESIGN.md sections 3 and 9): the flow
// field moved non-temporally (-1 %); the short reciprocal in the solve (-0.7 %); and three probes that locate the limit -- 16 vector
// instructions per step fewer in the producers: +-0; the fifth channel of phase B moved to two producer waves: +4.5 %; the same
// short form for the two sqrtf of the magnitude sum: -0.15 %.  Both roles are as long as the step, the producers paced by the
// bytes they pull through L1, the consumers by the length of their dependent chains.
struct FiVec {                  // a flow vector in flight: IN 0 uses p00 only
    float2 p00, p01, p10, p11;
    float b1;
};

#if IPX_FI_C_BPERMUTE
// the f64 value `v` of the lane whose byte address (4 * lane) is `addr`
__device__ __forceinline__ double fi_lane_f64(double v, int addr)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
#endif

// 1 / x for a normal x of modest exponent, correctly rounded: the steps of the compiler's IEEE division of 1.0 by x
// that remain when nothing needs scaling (see phase C)
__device__ __forceinline__ bool fi_is_normal(double x) { return __builtin_amdgcn_class(x, 0x108); }   // +-normal
__device__ __forceinline__ double fi_rcp_normal(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-x, y, 1.0);          // residual of the quotient q = 1.0 * y
    return __builtin_fma(e, y, y);
}

// workgroup barrier that orders LDS traffic only: global loads stay in flight across it
__device__ __forceinline__ void fi_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef IPX_FI_STAMPS
// Debug build only (python -m image_processing_amd.build --stamps): shader-clock totals per wave and phase
// of every 64th workgroup, read back with ipx_debug_fi_stamps().
__device__ unsigned long long fi_stamps[8][8];
__device__ unsigned long long fi_wg_times[2][16384];     // start / end (s_memtime) of every workgroup of the last level-0 launch
#define FI_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[k] += t_ - st_t; st_t = t_; } while (0)
#else
#define FI_STAMP(k) do { } while (0)
#endif

template <int IN, bool REDUCE>
__global__ __launch_bounds__(FI_THREADS, 4) void flow_iter_kernel(const FiArgs a)
{
    extern __shared__ __align__(16) unsigned char fi_lds[];
    float *s_m = reinterpret_cast<float *>(fi_lds);                                        // [slot][ch][c]
    double *s_v = reinterpret_cast<double *>(fi_lds + FI_RING * FI_MROW * sizeof(float));  // [ch][r][even | odd columns]
    const float *R = a.R;
    const int h = a.h, w = a.w, seg_rows = a.seg_rows, strips = a.strips, np = a.np, ppx = a.ppx;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.prio == 1 || (a.prio == 2 && wave < 4) || (a.prio == 3 && wave >= 4)) __builtin_amdgcn_s_setprio(3);
    // Workgroups go round-robin over the 8 XCDs (id % 8) and every XCD has its own L2.  XCD x gets the
    // pairs [x*ppx, (x+1)*ppx), and its resident workgroups (64: two per CU) are the same few strips of consecutive
    // pairs: the expansion of frame p+1 (sampled by pair p, read in place by pair p+1) is then fetched once per L2
    // instead of twice, and so are the 14 halo columns and the cache lines that two neighbouring strips share
    // (a strip starts at column 114 s - 7: not on a line boundary).
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    int pl = k % ppx, t = k / ppx;
    if (a.sgroup > 1) {
        // sgroup neighbouring strips of ppx / sgroup consecutive pairs resident together instead of one strip of ppx pairs:
        // the 14 halo columns between two strips of a group are L2 hits too, for an expansion more at the group's ends
        const int per_seg = ppx * strips, seg = k / per_seg, ks = k - seg * per_seg;
        const int gi = ks / (a.sgroup * ppx), rem = ks - gi * a.sgroup * ppx;
        const int gs = min(a.sgroup, strips - gi * a.sgroup);
        pl = rem / gs;
        t = seg * strips + gi * a.sgroup + (rem - pl * gs);
    }
    const int x0 = (t % strips) * FI_TW;
    const int ys = (t / strips) * seg_rows;
    const size_t p = (size_t)xcd * ppx + pl;
    if (ys >= h || p >= (size_t)np) return;
    const int ye = min(h, ys + seg_rows);
    const size_t P = (size_t)h * w;
    const unsigned FS = (unsigned)(r_stride(P) * 4), P16 = (unsigned)(P * 16);
    const __amdgpu_buffer_rsrc_t rR = ipx_make_rsrc(R + p * r_stride(P), 2 * FS);
    const float2 *fin = a.flow_in + p * (IN == 2 ? (size_t)a.sh * a.sw : P);
    float2 *fout = a.flow_out + p * P;

    const int ac = tid & (FI_MC - 1);                    // matrix column of phases A and B
    int ax = x0 - FI_M + ac;
    const bool col_used = ax <= w - 1 + FI_M;            // matrix columns further right feed no output of the image
    ax = ax < 0 ? 0 : (ax > w - 1 ? w - 1 : ax);
    auto clamp_row = [&](int j) { return j < 0 ? 0 : (j > h - 1 ? h - 1 : j); };
    // matrix row j (unclamped) lives in ring slot (j - ys + 8) % RING
    auto ring_store = [&](int j, const float (&m)[5]) {
        float *dst = s_m + ((j - ys + 8) % FI_RING) * FI_MROW + ac;
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) dst[ch * FI_MC] = m[ch];
    };
    // flow vector of pixel (ax, row): the loads (vec_issue) are kept in flight across most of a step
    // (the two horizontal taps are neighbours: one 16-byte load per coarse row; at the right edge, where the
    // reference uses the last sample alone, the pair ends at it)
    struct FiTaps { float2 l, r; };                     // 8-byte aligned
    int usx = 0; float ua1 = 0.f; bool uedge = false;
    if (IN == 2) {
        usx = a.x_ofs[ax]; ua1 = a.x_a[ax];
        uedge = usx >= a.sw - 1;
        usx = max(0, min(usx, a.sw - 2));
    }
    // The row tables of the upsampling are read one step before the flow loads that need them (row_tab, then
    // vec_issue_t), and as scalar loads: because the kernel also writes memory the compiler would otherwise
    // read them with vector loads, and a wave that waits for one of those cannot queue its next loads.
    struct FiRow { int row, o0, o1; float b1; };
    typedef const int __attribute__((address_space(4))) *fi_ctab_i;      // read-only for the whole launch:
    typedef const float __attribute__((address_space(4))) *fi_ctab_f;    // constant address space = scalar loads
    const fi_ctab_i c_yofs = (fi_ctab_i)(uintptr_t)a.y_ofs;
    const fi_ctab_f c_yb = (fi_ctab_f)(uintptr_t)a.y_b;
    auto row_tab = [&](int row) {
        FiRow t; t.row = row; t.o0 = t.o1 = 0; t.b1 = 0.f;
        if (IN == 2) { t.o0 = c_yofs[2 * row]; t.o1 = c_yofs[2 * row + 1]; t.b1 = c_yb[row]; }
        return t;
    };
    auto vec_issue_t = [&](const FiRow &t) {
        FiVec v;
        if (IN == 0) {
            typedef float fi_f2 __attribute__((ext_vector_type(2)));
#ifdef IPX_FI_FLOW_TEMPORAL   // (A/B build)
            const fi_f2 lv = *reinterpret_cast<const fi_f2 *>(fin + (size_t)t.row * w + ax);
#else
            const fi_f2 lv = __builtin_nontemporal_load(reinterpret_cast<const fi_f2 *>(fin + (size_t)t.row * w + ax));
#endif
            v.p00 = make_float2(lv.x, lv.y);
        } else if (IN == 2) {
            const float2 *S0 = fin + (size_t)t.o0 * a.sw, *S1 = fin + (size_t)t.o1 * a.sw;
            v.b1 = t.b1;
            const FiTaps t0 = *reinterpret_cast<const FiTaps *>(S0 + usx), t1 = *reinterpret_cast<const FiTaps *>(S1 + usx);
            v.p00 = t0.l; v.p01 = t0.r; v.p10 = t1.l; v.p11 = t1.r;
        }
        return v;
    };
    auto vec_issue = [&](int row) { return vec_issue_t(row_tab(row)); };
    auto vec_finish = [&](const FiVec &v) {
        if (IN == 0) {
            // An opaque copy: the loaded vector's register is free again here, in the middle of a step, so the
            // next load can target it.  Without it the value stays live into the step after next, the registers
            // rotate at the loop's end, and that copy waits for every load the wave has in flight.
            float2 o;
            asm("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(o.x), "=&v"(o.y) : "v"(v.p00.x), "v"(v.p00.y));
            return o;
        }
        if (IN == 1) return make_float2(0.f, 0.f);
        float2 h0 = v.p01, h1 = v.p11, o;
        if (!uedge) {
            const float a1 = ua1, a0 = 1.f - a1;
            h0.x = v.p00.x * a0 + v.p01.x * a1; h0.y = v.p00.y * a0 + v.p01.y * a1;
            h1.x = v.p10.x * a0 + v.p11.x * a1; h1.y = v.p10.y * a0 + v.p11.y * a1;
        }
        const float b1 = v.b1, b0 = 1.f - b1;
        o.x = (h0.x * b0 + h1.x * b1) * 2.0f;
        o.y = (h0.y * b0 + h1.y * b1) * 2.0f;
        return o;
    };

    // rows ys-7 .. ys+10 (everything step 0 reads), all waves: wave pair i fills rows ys-7+i, +4, ...
#pragma unroll 1
    for (int j = ys - FI_M + (wave >> 1); j <= ys + FI_M + FI_G - 1; j += FI_G) {
        const int y = clamp_row(j);
        const float2 d = vec_finish(vec_issue(y));
        UmIn in;
        float m[5];
        um_load(rR, FS, P16, d, ax, y, h, w, in, col_used);
        um_finish(in, d, ax, y, h, w, m);
        ring_store(j, m);
    }
    const bool producer = wave >= 4;
    const int prow = (wave - 4) >> 1;                    // producers: rows prow and prow + 2 of a group of 4
    FiVec dv0, dv1;
    if (producer) {
        dv0 = vec_issue(clamp_row(ys + FI_M + FI_G + prow));
        dv1 = vec_issue(clamp_row(ys + FI_M + FI_G + 2 + prow));
    }
    fi_barrier();

    // phase B items (consumer waves only): column ac, channels 0-2 on waves 0-1 and 3-4 on waves 2-3
    const int bhalf = (wave >> 1) & 1, bch0 = bhalf * 3, nbch = 3 - bhalf;
    const float *colb = s_m + bch0 * FI_MC + ac;
    const int vi = (ac >> 1) + (ac & 1) * FI_HALF;
    double *vob = s_v + bch0 * (FI_G * FI_MC) + vi;
    double vs[3] = {0.0, 0.0, 0.0};
    // The reference's vertical sum is a running sum from the top of the image whose every update is rounded
    // to f32 first: vsum = f32(row0 * 9) + rows 1..6, then per row vsum += f32(row[y+7] - row[y-8]).  A segment
    // that starts at row 0 repeats exactly that (run_pairs uses one segment per strip); a segment that starts
    // further down would begin with the plain sum of its 15 rows, which differs in the last bits.
    const bool from_top = ys == 0;
    if (!producer) {
        if (from_top) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < nbch) vs[c] = (double)(colb[8 * FI_MROW + c * FI_MC] * (float)(FI_M + 2));   // slot 8 = row 0
        }
        for (int i = from_top ? 9 : 1; i <= 2 * FI_M; ++i) {   // rows ys-7 .. ys+6 = slots 1 .. 14
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < nbch) vs[c] += (double)colb[i * FI_MROW + c * FI_MC];
        }
    }
#if IPX_FI_C_BPERMUTE
    const int lane_a1 = ((tid + 1) & 63) << 2, lane_a3 = ((tid + 3) & 63) << 2, lane_a5 = ((tid + 5) & 63) << 2,
              lane_a7 = ((tid + 7) & 63) << 2;             // phase C: the lanes 1, 3, 5, 7 to the right (ds_bpermute addresses)
#endif
    int slot_in = 15;                                    // slot of row yb+7 for yb = ys
    int slot_out = 1;                                    // slot of row ys-7, the first row to leave the window
    double msum = 0.0;
    unsigned *s_flag = reinterpret_cast<unsigned *>(fi_lds + FI_LDS_BYTES - 16);   // B -> C hand-off counter of the consumers
    if (tid == 0) *s_flag = 0;
    unsigned flag_target = 0;
    fi_barrier();

    auto phase_b = [&](bool first_step) {
#pragma unroll
        for (int r = 0; r < FI_G; ++r) {
            const float *pin = colb + slot_in * FI_MROW, *pout = colb + slot_out * FI_MROW;
            if (first_step && r == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < nbch) vs[c] += from_top ? (double)(pin[c * FI_MC] - colb[FI_MROW + c * FI_MC])   // row -8 = row 0 = slot 1
                                                    : (double)pin[c * FI_MC];
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if (c < nbch) vs[c] += (double)(pin[c * FI_MC] - pout[c * FI_MC]);
                slot_out = slot_out + 1 == FI_RING ? 0 : slot_out + 1;
            }
            slot_in = slot_in + 1 == FI_RING ? 0 : slot_in + 1;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c < nbch) vob[c * (FI_G * FI_MC) + r * FI_MC] = vs[c];
        }
    };
    // Every step but the first: the same update without a branch, all of the step's ring reads issued before the
    // first is used (rolled up row by row the wave would pay the LDS round trip four times per step).
    auto phase_b_fast = [&](auto nch_tag) {
        constexpr int NCH = decltype(nch_tag)::value;
        float din[FI_G][NCH], dout[FI_G][NCH];
#pragma unroll
        for (int r = 0; r < FI_G; ++r) {
            const float *pin = colb + slot_in * FI_MROW, *pout = colb + slot_out * FI_MROW;
#pragma unroll
            for (int c = 0; c < NCH; ++c) { din[r][c] = pin[c * FI_MC]; dout[r][c] = pout[c * FI_MC]; }
            slot_out = slot_out + 1 == FI_RING ? 0 : slot_out + 1;
            slot_in = slot_in + 1 == FI_RING ? 0 : slot_in + 1;
        }
#pragma unroll
        for (int r = 0; r < FI_G; ++r)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                vs[c] += (double)(din[r][c] - dout[r][c]);
                vob[c * (FI_G * FI_MC) + r * FI_MC] = vs[c];
            }
    };
    // the four consumer waves meet here between B and C; the producers are not held up by it
    auto consumers_sync = [&]() {
        flag_target += 4;
        if ((tid & 63) == 0) __hip_atomic_fetch_add(s_flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // every consumer wave posts once per step, so the wait ends; the bound only keeps a future protocol
        // error from hanging the GPU (~10 s of polling at the default limit).  Running into it is not silent:
        // the stage's error word is raised, magnitude_finish_kernel then reports NaN metrics with keep = 0 and
        // ipx_synchronize / ipx_flow_status return IPX_ERR_HIP.
        unsigned spins = 0;
        bool met;
        while (!(met = __hip_atomic_load(s_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= flag_target) && spins < a.spin_limit) {
            ++spins;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!met && (tid & 63) == 0) atomicOr(a.err, 1u);
    };

#ifdef IPX_FI_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime();
    if (tid == 0 && h > 600 && blockIdx.x < 16384) fi_wg_times[0][blockIdx.x] = st_t;
#endif
    if (producer) {
        // Software pipeline over matrix rows: while one of the thread's two rows is finished and stored, the
        // loads of the other (and of the row that replaces the finished one) are in flight, so the CU's
        // memory pipe always has this wave's requests queued.
        UmIn in0, in1;
        float2 d0 = vec_finish(dv0), d1 = vec_finish(dv1);
        FiRow tb0, tb1;
        {
            const int j0 = ys + FI_M + FI_G + prow, j1 = j0 + 2;     // rows of step 1
            const bool more = ys + FI_G < ye;
            um_load(rR, FS, P16, d0, ax, clamp_row(j0), h, w, in0, more && col_used);
            um_load(rR, FS, P16, d1, ax, clamp_row(j1), h, w, in1, more && col_used);
            dv0 = vec_issue(clamp_row(j0 + FI_G));
            dv1 = vec_issue(clamp_row(j1 + FI_G));
            tb0 = row_tab(clamp_row(j0 + 2 * FI_G));
            tb1 = row_tab(clamp_row(j1 + 2 * FI_G));
        }
        // s_waitcnt vmcnt(0).  The compiler issues the loads above in an order of its own, and the counts it
        // then waits for inside the loop are the worst case over both ways into it: without an empty queue
        // here every step would begin by waiting for all of the wave's loads, the row in flight included.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // (rows of the loop are >= ys + 11: only the bottom clamp applies; the ring slots of the thread's two rows
        // advance by 4 per step -- both spare the scalar unit a division and two selects per use)
        auto clamp_hi = [&](int j) { return j > h - 1 ? h - 1 : j; };
        auto ring_put = [&](int slot, const float (&m)[5]) {
            float *dst = s_m + slot * FI_MROW + ac;
#pragma unroll
            for (int ch = 0; ch < 5; ++ch) dst[ch * FI_MC] = m[ch];
        };
        int slot0 = (FI_M + FI_G + prow + 8) % FI_RING, slot1 = (FI_M + FI_G + prow + 10) % FI_RING;
#pragma unroll 1
        for (int yb = ys; yb < ye; yb += FI_G) {
            const bool more = yb + FI_G < ye;            // the next step needs matrix rows yb+11 .. yb+14 (in flight)
            const bool more2 = yb + 2 * FI_G < ye;       // and the one after it rows yb+15 .. yb+18 (issued here)
            const int j0 = yb + FI_M + FI_G + prow, j1 = j0 + 2;
            float m[5];
            um_finish(in0, d0, ax, clamp_hi(j0), h, w, m);
            if (more) ring_put(slot0, m);
            slot0 = slot0 + FI_G >= FI_RING ? slot0 + FI_G - FI_RING : slot0 + FI_G;
            FI_STAMP(0);
            d0 = vec_finish(dv0);
            um_load(rR, FS, P16, d0, ax, clamp_hi(j0 + FI_G), h, w, in0, more2 && col_used);
            dv0 = vec_issue_t(tb0);
            tb0 = row_tab(clamp_hi(j0 + 3 * FI_G));
            FI_STAMP(1);
            um_finish(in1, d1, ax, clamp_hi(j1), h, w, m);
            if (more) ring_put(slot1, m);
            slot1 = slot1 + FI_G >= FI_RING ? slot1 + FI_G - FI_RING : slot1 + FI_G;
            FI_STAMP(2);
            d1 = vec_finish(dv1);
            um_load(rR, FS, P16, d1, ax, clamp_hi(j1 + FI_G), h, w, in1, more2 && col_used);
            dv1 = vec_issue_t(tb1);
            tb1 = row_tab(clamp_hi(j1 + 3 * FI_G));
            FI_STAMP(3);
            fi_barrier();                                // ring rows of the next step written
            FI_STAMP(4);
        }
    } else {
#pragma unroll 1
        for (int yb = ys; yb < ye; yb += FI_G) {
#if IPX_FI_ABLATE & 1
            fi_barrier();
            continue;
#endif
            if (yb == ys) phase_b(true);
            else if (bhalf == 0) phase_b_fast(std::integral_constant<int, 3>());
            else phase_b_fast(std::integral_constant<int, 2>());
            FI_STAMP(1);
            consumers_sync();
            FI_STAMP(2);
            // ---- C: horizontal sums + solve, row r = wave, outputs 2l, 2l+1 (l < 57)
            const int r = wave, l = tid & 63;
            const int y = yb + r;
            if (y < ye) {
                // Window of output 2l = matrix columns 2l .. 2l+14, of output 2l+1 = columns 2l+1 .. 2l+15; even
                // columns at [l + i], odd at [HALF + l + i].  The 15-sums are built from sums of column pairs and
                // quads that the lanes of the row's wave hand each other through the LDS crossbar (ds_bpermute: no
                // LDS storage), 8 f64 additions and 13 LDS-pipe operations per channel instead of 16 and 17:
                //   p[l] = v[2l] + v[2l+1],  q[l] = p[l] + p[l+1],  Q = q[l+1] + q[l+3] + q[l+5]  (columns 2l+2 .. 2l+13)
                //   window(2l) = (p[l] + Q) + v[2l+14],  window(2l+1) = (v[2l+1] + Q) + p[l+7]
                // All 64 lanes take part (lanes 57-63 only as sources; what they compute themselves is dropped).
                double acc[2][5];
#if IPX_FI_C_BPERMUTE
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) {
                    const double *ve = s_v + ch * (FI_G * FI_MC) + r * FI_MC + l;
                    const double v0 = ve[0], v1 = ve[FI_HALF], v14 = ve[FI_M];
                    const double pr = v0 + v1;
                    const double q = pr + fi_lane_f64(pr, lane_a1);
                    const double Q = (fi_lane_f64(q, lane_a1) + fi_lane_f64(q, lane_a3)) + fi_lane_f64(q, lane_a5);
                    acc[0][ch] = (pr + Q) + v14;
                    acc[1][ch] = (v1 + Q) + fi_lane_f64(pr, lane_a7);
                }
#else
                // The lanes hand each other p and q through the row's own LDS words, which only this wave touches
                // and whose values are in registers by then: p[l] replaces v[2l], q[l] replaces v[2l+1].  A wave's
                // LDS operations execute in order, so a later read sees an earlier write of another lane; the reads
                // are volatile asm statements with a memory clobber, which keeps the compiler from moving the write of
                // one word across the read of another (to one thread they are unrelated).  8-byte reads and writes
                // cost the LDS pipe 2 and 6 cycles, the two ds_bpermute of an f64 exchange 4 each: 28 instead of 46
                // cycles per channel.
                // The reads are spelled out as ds_read_b64: the compiler would pair them into ds_read2_b64, which takes
                // the LDS array 8 cycles for two values where two ds_read_b64 take 4 (MI355X_MICROARCH.md, LDS table).
                // Byte offsets: channel 4096 (FI_G * FI_MC doubles), odd columns 512 (FI_HALF), 7 lanes on 56 (FI_M).
                static_assert(FI_G * FI_MC * 8 == 4096 && FI_HALF * 8 == 512 && FI_M * 8 == 56, "offsets of the ds_read_b64 blocks");
                double *vrow = s_v + r * FI_MC + l;
                const unsigned va = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double *)vrow;
                const unsigned vqa = va + 512u - 8u * (unsigned)(l - min(l, FI_HALF - 6));   // lanes 59-63 own no output: keep their reads inside the row
                double v0[5], v1[5], v14[5], pr[5], q[5];
                asm volatile("ds_read_b64 %0, %15\n\tds_read_b64 %1, %15 offset:512\n\tds_read_b64 %2, %15 offset:56\n\t"
                             "ds_read_b64 %3, %15 offset:4096\n\tds_read_b64 %4, %15 offset:4608\n\tds_read_b64 %5, %15 offset:4152\n\t"
                             "ds_read_b64 %6, %15 offset:8192\n\tds_read_b64 %7, %15 offset:8704\n\tds_read_b64 %8, %15 offset:8248\n\t"
                             "ds_read_b64 %9, %15 offset:12288\n\tds_read_b64 %10, %15 offset:12800\n\tds_read_b64 %11, %15 offset:12344\n\t"
                             "ds_read_b64 %12, %15 offset:16384\n\tds_read_b64 %13, %15 offset:16896\n\tds_read_b64 %14, %15 offset:16440\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0[0]), "=&v"(v1[0]), "=&v"(v14[0]), "=&v"(v0[1]), "=&v"(v1[1]), "=&v"(v14[1]), "=&v"(v0[2]), "=&v"(v1[2]),
                               "=&v"(v14[2]), "=&v"(v0[3]), "=&v"(v1[3]), "=&v"(v14[3]), "=&v"(v0[4]), "=&v"(v1[4]), "=&v"(v14[4])
                             : "v"(va) : "memory");
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) pr[ch] = v0[ch] + v1[ch];
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) vrow[ch * (FI_G * FI_MC)] = pr[ch];
                double p1[5], p7[5];
                asm volatile("ds_read_b64 %0, %10 offset:8\n\tds_read_b64 %1, %10 offset:56\n\t"
                             "ds_read_b64 %2, %10 offset:4104\n\tds_read_b64 %3, %10 offset:4152\n\t"
                             "ds_read_b64 %4, %10 offset:8200\n\tds_read_b64 %5, %10 offset:8248\n\t"
                             "ds_read_b64 %6, %10 offset:12296\n\tds_read_b64 %7, %10 offset:12344\n\t"
                             "ds_read_b64 %8, %10 offset:16392\n\tds_read_b64 %9, %10 offset:16440\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(p1[0]), "=&v"(p7[0]), "=&v"(p1[1]), "=&v"(p7[1]), "=&v"(p1[2]), "=&v"(p7[2]), "=&v"(p1[3]), "=&v"(p7[3]),
                               "=&v"(p1[4]), "=&v"(p7[4])
                             : "v"(va) : "memory");
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) q[ch] = pr[ch] + p1[ch];
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) vrow[ch * (FI_G * FI_MC) + FI_HALF] = q[ch];
                double q1[5], q3[5], q5[5];
                asm volatile("ds_read_b64 %0, %15 offset:8\n\tds_read_b64 %1, %15 offset:24\n\tds_read_b64 %2, %15 offset:40\n\t"
                             "ds_read_b64 %3, %15 offset:4104\n\tds_read_b64 %4, %15 offset:4120\n\tds_read_b64 %5, %15 offset:4136\n\t"
                             "ds_read_b64 %6, %15 offset:8200\n\tds_read_b64 %7, %15 offset:8216\n\tds_read_b64 %8, %15 offset:8232\n\t"
                             "ds_read_b64 %9, %15 offset:12296\n\tds_read_b64 %10, %15 offset:12312\n\tds_read_b64 %11, %15 offset:12328\n\t"
                             "ds_read_b64 %12, %15 offset:16392\n\tds_read_b64 %13, %15 offset:16408\n\tds_read_b64 %14, %15 offset:16424\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(q1[0]), "=&v"(q3[0]), "=&v"(q5[0]), "=&v"(q1[1]), "=&v"(q3[1]), "=&v"(q5[1]), "=&v"(q1[2]), "=&v"(q3[2]),
                               "=&v"(q5[2]), "=&v"(q1[3]), "=&v"(q3[3]), "=&v"(q5[3]), "=&v"(q1[4]), "=&v"(q3[4]), "=&v"(q5[4])
                             : "v"(vqa) : "memory");
#pragma unroll
                for (int ch = 0; ch < 5; ++ch) {
                    const double Q = (q1[ch] + q3[ch]) + q5[ch];
                    acc[0][ch] = (pr[ch] + Q) + v14[ch];
                    acc[1][ch] = (v1[ch] + Q) + p7[ch];
                }
#endif
                const double scale = 1.0 / (double)((2 * FI_M + 1) * (2 * FI_M + 1));
                float2 o[2];
                double g[2][5], det[2], idet[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int ch = 0; ch < 5; ++ch) g[j][ch] = acc[j][ch] * scale;
                    det[j] = g[j][0] * g[j][2] - g[j][1] * g[j][1] + 1e-3;
                }
                // 1 / det.  The determinant is built from f32 matrix entries (|det| < 2^273) and is almost always a normal
                // number; for such a divisor the IEEE division the compiler emits (v_div_scale, v_rcp, two Newton steps,
                // quotient, residual, v_div_fmas, v_div_fixup: 11 instructions) scales nothing and fixes nothing up, and
                // fi_rcp_normal is the same arithmetic without those steps (7 instructions, the same bits).  A wave in
                // which some lane holds a zero, a denormal, an infinity or a NaN takes the division.
                if (__builtin_amdgcn_ballot_w64(!(fi_is_normal(det[0]) && fi_is_normal(det[1]))) == 0) {
                    idet[0] = fi_rcp_normal(det[0]); idet[1] = fi_rcp_normal(det[1]);
                } else {
                    idet[0] = 1. / det[0]; idet[1] = 1. / det[1];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    o[j].x = (float)((g[j][0] * g[j][4] - g[j][1] * g[j][3]) * idet[j]);
                    o[j].y = (float)((g[j][2] * g[j][3] - g[j][1] * g[j][4]) * idet[j]);
                }
                const int x = l < FI_TW / 2 ? x0 + 2 * l : w;     // lanes 57-63 own no output
                // both solves are finished before anything is stored: left to itself the compiler moves the second one
                // behind the test of x + 1 < w, and the two division chains then run one after the other
                asm volatile("" : "+v"(o[0].x), "+v"(o[0].y), "+v"(o[1].x), "+v"(o[1].y));
                if (REDUCE) {
                    float m0 = sqrtf(o[0].x * o[0].x + o[0].y * o[0].y), m1 = sqrtf(o[1].x * o[1].x + o[1].y * o[1].y);   // cartToPolar magnitude in f32
                    asm volatile("" : "+v"(m0), "+v"(m1));
                    msum += (double)(x < w ? m0 : 0.f);           // (adding +0 changes nothing: the sum is never -0)
                    msum += (double)(x + 1 < w ? m1 : 0.f);
                } else {
                    // one 16-byte store per lane where the width is even (x is even: x < w then means x + 1 < w, and the
                    // address is 16-byte aligned).  Under the test x + 1 < w the compiler made two 8-byte stores of it.
                    float2 *dst = fout + (size_t)y * w + x;
                    if ((w & 1) == 0) {
                        if (x < w) {
                            // (non-temporal, like the load of the field in vec_issue_t: an iteration's field is read by the
                            // next one 10 ms later, by which time no cache holds it; marked so, it leaves the L2 to the
                            // expansions, each of which a second pair reads: -1 % on the kernel, measured A/B)
                            typedef float fi_f4 __attribute__((ext_vector_type(4)));
                            const fi_f4 ov = {o[0].x, o[0].y, o[1].x, o[1].y};
#ifdef IPX_FI_STORE_TEMPORAL   // (A/B build)
                            *reinterpret_cast<fi_f4 *>(dst) = ov;
#else
                            __builtin_nontemporal_store(ov, reinterpret_cast<fi_f4 *>(dst));
#endif
                        }
                    } else {
                        if (x < w) dst[0] = o[0];
                        if (x + 1 < w) dst[1] = o[1];
                    }
                }
            }
            FI_STAMP(3);
            fi_barrier();                                // s_v free again
            FI_STAMP(4);
        }
    }
#ifdef IPX_FI_STAMPS
    if (tid == 0 && h > 600 && blockIdx.x < 16384) fi_wg_times[1][blockIdx.x] = __builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0 && (blockIdx.x & 63) == 0 && h > 600) {
        for (int i = 0; i < 8; ++i) atomicAdd(&fi_stamps[wave][i], st_acc[i]);
        atomicAdd(&fi_stamps[wave][7], (unsigned long long)((ye - ys + FI_G - 1) / FI_G));
    }
#endif
    if (REDUCE) {                                        // fixed-order sum of the workgroup's magnitudes
        for (int off = 32; off > 0; off >>= 1) msum += __shfl_down(msum, off);
        if ((tid & 63) == 0) s_v[wave] = msum;
        fi_barrier();
        if (tid == 0) a.partial[p * (size_t)(strips * a.nseg) + t] = (s_v[0] + s_v[1]) + (s_v[2] + s_v[3]);
    }
}

}  // namespace

void ipx_flow_iter_launch(hipStream_t stream, int in_mode, bool reduce, dim3 grid, const FiArgs &a)
{
    const dim3 block(FI_THREADS);
    if (in_mode == 0 && reduce) hipLaunchKernelGGL((flow_iter_kernel<0, true>), grid, block, FI_LDS_BYTES, stream, a);
    else if (in_mode == 0) hipLaunchKernelGGL((flow_iter_kernel<0, false>), grid, block, FI_LDS_BYTES, stream, a);
    else if (in_mode == 1) hipLaunchKernelGGL((flow_iter_kernel<1, false>), grid, block, FI_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((flow_iter_kernel<2, false>), grid, block, FI_LDS_BYTES, stream, a);
}

// the marching kernel's LDS (ring + sums) is above the 64 KB a kernel gets without asking
hipError_t ipx_flow_iter_prepare()
{
    hipError_t e = hipFuncSetAttribute((const void *)flow_iter_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FI_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)flow_iter_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FI_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)flow_iter_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FI_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)flow_iter_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FI_LDS_BYTES);
    return e;
}

#ifdef IPX_FI_STAMPS
extern "C" __attribute__((visibility("default"))) int ipx_debug_fi_wg_times(unsigned long long *out, int n)
{
    if (n > 16384) n = 16384;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fi_wg_times), sizeof(unsigned long long) * n, 0) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out + n, HIP_SYMBOL(fi_wg_times), sizeof(unsigned long long) * n, sizeof(unsigned long long) * 16384) != hipSuccess) return -1;
    return 0;
}
extern "C" __attribute__((visibility("default"))) int ipx_debug_fi_stamps(unsigned long long *out64, int reset)
{
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(fi_stamps), sizeof(unsigned long long) * 64) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(fi_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

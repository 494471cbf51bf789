// Internal declarations shared by the HIP translation units of libipx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include "../../include/ipx.h"

// Scratch arenas of a context.  Each slot is an independent growable device buffer; a stage
// only touches the slots listed next to it, so the chunk orchestrator can keep its planes in
// the WS_* slots while the stages it calls reuse theirs.
enum ipx_slot {
    SLOT_CCL = 0,        // ccl.hip: parent/aux arrays + one plane per batched frame
    SLOT_STAGE,          // single-stage entry points (ipx_background's dummy plane, ...)
    SLOT_FLOW_TAB,       // flow.hip: per-(H,W) resize / filter tables
    SLOT_FLOW_PYR,       // flow.hip: level images + polynomial expansions of a batch
    SLOT_FLOW_UV,        // flow.hip: flow fields (two levels, ping-pong)
    SLOT_FLOW_RED,       // flow.hip: partial sums of the magnitude reduction
    WS_RAW, WS_BRIGHT, WS_PERS, WS_STATIC, WS_FINAL, WS_MAPS, WS_COUNTS,   // chunk.hip (and the ranks of group.hip)
    GS_FRAMES, GS_FRAMES_B, GS_MISC, GS_EDGES, GS_PERS_ALL, GS_MAPS_ALL,   // group.hip: a rank's frames (two buffers: the next chunk uploads under this one), small arrays, exchange areas
    SLOT_COUNT
};

struct ipx_flow_plan;   // flow.hip

// Kernel classes of the built-in profiler (ipx_profile_*): HIP events recorded on the context's
// stream around every launch, accumulated per class together with the launch's ALGORITHMIC bytes
// (compulsory reads + writes: every input and output array counted once; DESIGN.md lists them).
enum ipx_kid {
    KID_PYR = 0, KID_POLYEXP, KID_FLOW_ITER,
    KID_DETECT, KID_DAMAGED_MASK,
    KID_CCL_INIT, KID_CCL_MERGE, KID_CCL_REDUCE, KID_CCL_SELECT, KID_CLOSE, KID_CLUSTER_EVAL,
    KID_PERSISTENT, KID_STATIC_MAPS, KID_STATIC_FROM_MAPS, KID_FINALIZE, KID_MISC,
    KID_COUNT
};

#define IPX_PROF_LEVELS 8
struct ipx_prof_rec { int kid, level; double bytes; hipEvent_t e0, e1; };
struct ipx_prof {
    int on;
    ipx_prof_rec *recs; int nrecs, cap;     // pending (not yet folded) launches; events are reused
    double ms[KID_COUNT], bytes[KID_COUNT];
    unsigned long long launches[KID_COUNT];
    // the flow_iter class once more, by pyramid level of the launch (ipx_profile_read_flow_level)
    double lvl_ms[IPX_PROF_LEVELS], lvl_bytes[IPX_PROF_LEVELS];
    unsigned long long lvl_launches[IPX_PROF_LEVELS];
};

struct ipx_ctx {
    int device;
    hipStream_t stream;
    ipx_params p;
    void *slot[SLOT_COUNT];
    size_t slot_bytes[SLOT_COUNT];
    hipEvent_t ev0, ev1;
    hipStream_t copy_stream;    // ipx_upload_frames: made on first use
    hipEvent_t copy_done;
    ipx_flow_plan *flow_plan;   // cached tables for the last (H,W) seen by the flow stage
    unsigned fi_spin_limit;     // flow_iter_kernel: polls of the consumers' hand-off before it is given up (error word)
    int flow_ran;               // the flow stage has used SLOT_FLOW_RED (whose first word is its error word)
    ipx_prof prof;
};

// roctx ranges: host-side markers around the enqueue of each stage of the path, so that a
// rocprofv3 --marker-trace run carries the stage names beside the kernel trace.  libroctx64.so is resolved with
// dlopen at first use (no link-time dependency); without it, or with IPX_ROCTX=0, the calls do nothing.
void ipx_range_push(const char *name);
void ipx_range_pop(void);
struct ipx_range {
    explicit ipx_range(const char *name) { ipx_range_push(name); }
    ~ipx_range() { ipx_range_pop(); }
    ipx_range(const ipx_range &) = delete;
    ipx_range &operator=(const ipx_range &) = delete;
};

void ipx_prof_begin(ipx_ctx *ctx, int kid, double bytes, int level = -1);   // level: pyramid level of a flow_iter launch
void ipx_prof_end(ipx_ctx *ctx);

// Launch `kernel` on the context's stream; `abytes` = algorithmic bytes of this launch.
#define IPX_LAUNCH(ctx, kid, abytes, kernel, grid, block, ...)                          \
    do {                                                                               \
        if ((ctx)->prof.on) ipx_prof_begin((ctx), (kid), (double)(abytes));            \
        hipLaunchKernelGGL(kernel, grid, block, 0, (ctx)->stream, __VA_ARGS__);        \
        if ((ctx)->prof.on) ipx_prof_end(ctx);                                         \
    } while (0)

// Same with `lds` bytes of dynamic LDS.
#define IPX_LAUNCH_LDS(ctx, kid, abytes, lds, kernel, grid, block, ...)                 \
    do {                                                                               \
        if ((ctx)->prof.on) ipx_prof_begin((ctx), (kid), (double)(abytes));            \
        hipLaunchKernelGGL(kernel, grid, block, (lds), (ctx)->stream, __VA_ARGS__);    \
        if ((ctx)->prof.on) ipx_prof_end(ctx);                                         \
    } while (0)

// gfx950 + hipcc 7.2 (measured, tools/hwtest/seg_loop.hip): a 64-bit vector shift whose amount operand is the
// last VGPR of the wave's allocation sometimes shifts by a neighbouring wave's register.  A kernel that
// tools/check_shift64.py reports gets IPX_VGPR_PAD(n) with n = its next_free_vgpr: register v<n> counts as used,
// so the allocation grows by a granule and the shift operand is no longer the last register.  The check runs
// at every build.
#define IPX_VGPR_PAD(n) asm volatile("; v" #n " kept allocated (tools/check_shift64.py)" ::: "v" #n)
// The register the three affected kernels keep allocated: v62 makes next_free_vgpr 63, so the wave is given 64
// registers (still 8 waves per SIMD) of which the last, v63, is never handed out by the compiler: whatever
// register a shift amount lives in, it is not the last one of the allocation.  Independent of how many registers
// the kernel's code needs (as long as that is fewer than 62; the build's ISA scan fails otherwise).
#define IPX_VGPR_PAD_SAFE() IPX_VGPR_PAD(62)

void ipx_set_error(const char *fmt, ...);

#define IPX_CHECK_HIP(expr)                                                            \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            ipx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                          __FILE__, __LINE__);                                         \
            return IPX_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define IPX_REQUIRE(cond, msg)                                                         \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            ipx_set_error("%s: %s", __func__, msg);                                    \
            return IPX_ERR_ARG;                                                        \
        }                                                                              \
    } while (0)

#define IPX_TRY(expr)                                                                  \
    do {                                                                               \
        int _s = (expr);                                                               \
        if (_s != IPX_OK) return _s;                                                   \
    } while (0)

#define IPX_LAUNCH_CHECK() IPX_CHECK_HIP(hipGetLastError())

// Makes ctx->device current.
int ipx_ctx_enter(ipx_ctx *ctx);
// Grows slot `s` to at least `bytes` (synchronises the stream before freeing the old arena).
int ipx_scratch(ipx_ctx *ctx, int s, size_t bytes, void **out);
void ipx_flow_plan_free(ipx_flow_plan *plan);
// ccl.hip: resolves the device address of the static scratch of ipx_remove_bright_regions_noalloc on `device`
// (which must be current); ipx_ctx_create calls it.
int ipx_static_scratch_init(int device);

static inline int ipx_wpr(int W) { return (W + 63) >> 6; }
static inline size_t ipx_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Thresholds derived from ipx_params the way numpy compares (see include/ipx.h).
struct ipx_derived {
    int frame_gt;        // frame > brightness_threshold  <=>  (int)frame > frame_gt
    float bg_gt;         // background(f32) > (float)brightness_threshold
    int est_lt;          // frame < estimate_threshold
};
ipx_derived ipx_derive(const ipx_params &p);

// group.hip -- one chunk sharded by frame range across the GPUs of a node, inside the library:
// one context, one stream and one host thread per device (ipx_group_create /
// ipx_group_detect_chunk; section 8(e): the frame-range sharding of BASELINE configs[2]).
//
// The reference is a single process (ip.py:1204-1227 walks the chunks one after the other); a chunk's frames
// are independent except for three things, which are the only exchanges made here (the same three as
// image_processing_amd/sharding.py, which does this with one PROCESS per GPU over torch.distributed):
//   1. the background window (ip.py:189-190) and the flow of a shard's first frame (ip.py:856) read up to
//      `radius` frames of the neighbouring shards: every rank UPLOADS those halo frames with its own range;
//   2. persistence (ip.py:270-289) is a run length over the whole chunk: every rank evaluates its own raw planes
//      extended by the T-1 edge planes of its two neighbours and the ranks OR their maps;
//   3. the static map (ip.py:438-465) needs chunk-wide heat / valid sums: one sum of two int32 H x W maps.
// Two transports for 2 and 3, chosen at ipx_group_create:
//   IPX_GROUP_RCCL  ncclAllGather / ncclAllReduce on a communicator made by ncclCommInitAll (RCCL over xGMI);
//                   librccl is loaded with dlopen so that libipx.so itself does not depend on it;
//   IPX_GROUP_PEER  hipMemcpyPeerAsync from the neighbours' buffers, ordered by events and host barriers; also
//                   works when a device is listed more than once (how the sharding logic is tested on one GPU).
// Per-frame results go straight from every rank into the caller's HOST arrays (no collective).
//
// Two chunks may be in flight (ipx_group_submit / ipx_group_wait): every rank has two frame buffers and a side
// stream, and a chunk's frames are uploaded AT SUBMIT TIME, by the submitting thread, into the buffer the chunk
// before it does not use -- so the upload of chunk k+1 runs under the compute of chunk k (at 8 ranks x 125 frames of
// 1080p a rank uploads 0.27 GB for ~15 ms of compute: without the overlap the path is PCIe-bound by construction).
// The ranks' host threads take the chunks strictly in order; ipx_group_detect_chunk = submit + wait.
#include "ipx_internal.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// chunk.hip
int ipx_merge_counts(ipx_ctx *ctx, const int64_t *fin3, const int64_t *dense2, int n, int64_t *out5);

namespace {

// ---- the few RCCL entry points used, resolved at run time (types as in rccl.h) ---------------------------------
typedef struct ncclComm *ncclComm_t;
enum { NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_UINT64 = 5 };      // ncclDataType_t
enum { NCCL_SUM = 0 };                                          // ncclRedOp_t
struct Rccl {
    void *handle = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};

bool rccl_load(Rccl &r)
{
    static const char *const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return false;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.handle, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");      // optional: only used after a failure
    r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    return r.CommInitAll && r.CommDestroy && r.AllGather && r.AllReduce && r.GetErrorString;
}

// Reusable barrier of the ranks' host threads.  wait() also hands every rank the SAME view of the group's failure
// flag (sampled once, by the last rank to arrive): ranks decide together whether to go on, so none is left
// waiting at a later barrier for one that has gone.
struct HostBarrier {
    std::mutex m; std::condition_variable cv; int count = 0, waiting = 0; unsigned long gen = 0;
    bool verdict = false;
    bool wait(const std::atomic<bool> &failed)
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long g = gen;
        if (++waiting == count) { waiting = 0; verdict = failed.load(); ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
        return verdict;
    }
};

// dst[i] = sum over `count` stacked int32 arrays of n elements (the chunk-wide heat / valid maps)
__global__ void sum_i32_kernel(const int32_t *stack, int count, size_t n, int32_t *dst)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t s = 0;
    for (int k = 0; k < count; ++k) s += stack[(size_t)k * n + i];
    dst[i] = s;
}

struct Rank {
    int device = 0;
    ipx_ctx *ctx = nullptr;
    hipStream_t stream = nullptr, up_stream = nullptr;
    hipEvent_t ev_raw = nullptr, ev_pers = nullptr, ev_maps = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr};   // frames of the chunk in frame buffer 0 / 1 have arrived
    uint8_t *ext[2] = {nullptr, nullptr};       // this rank's frames (+ halo) of the chunk in that buffer
    double *tab[2] = {nullptr, nullptr};        // the caller's pow2 table, uploaded with them (behind the frames in the same slot)
    ncclComm_t comm = nullptr;
    // what the neighbours read (published before the barrier that precedes the read)
    uint64_t *raw = nullptr;       // own raw planes [nown]
    uint64_t *pers = nullptr;      // own persistence map
    int32_t *maps = nullptr;       // own heat | valid
    int nown = 0;
    int status = IPX_OK;
    char error[512] = "";
};

}  // namespace

struct GroupCall;

struct ipx_group {
    int nd = 0, transport = IPX_GROUP_PEER;
    std::vector<Rank> ranks;
    Rccl rccl;
    HostBarrier barrier;
    ipx_params p;
    // chunks in flight, oldest first (at most two); the ranks' threads take them strictly in order
    std::mutex m;
    std::condition_variable cv;
    std::deque<GroupCall *> inflight;
    unsigned long submitted = 0, completed = 0;
    bool broken = false;                       // an RCCL collective was abandoned: the communicator is gone
};

namespace {

inline void shard_range(int n, int r, int nd, int &a, int &b)
{
    a = (int)(((long long)r * n) / nd);
    b = (int)(((long long)(r + 1) * n) / nd);
}

}  // namespace

// One submitted chunk.
struct GroupCall {
    ipx_group *g;
    const uint8_t *frames; int n, H, W;
    std::vector<double> tab; int tdim;         // the caller's pow2 table, copied at submit
    float *flows; uint8_t *keep; int64_t *counts; int64_t *cluster;
    ipx_params p;
    int buf;                                   // frame buffer / upload event of this chunk (0 / 1)
    unsigned long seq;
    std::atomic<bool> failed{false};
    std::atomic<int> ranks_left{0};
    std::vector<std::thread> th;
    std::vector<int> status;                   // per rank, final (the ranks' own fields are reused by the next chunk)
    std::vector<std::string> error;
};
typedef GroupCall Call;

namespace {

void rank_fail(Call &c, Rank &rk, int status)
{
    rk.status = status;
    snprintf(rk.error, sizeof(rk.error), "%s", ipx_last_error());
    c.failed.store(true);   // only ever set during the call; the ranks read it through the next barrier
}

#define RK_TRY(expr)                                                                     \
    do {                                                                                 \
        if (rk.status == IPX_OK) { const int s_ = (expr); if (s_ != IPX_OK) rank_fail(c, rk, s_); } \
    } while (0)
#define RK_HIP(expr)                                                                     \
    do {                                                                                 \
        if (rk.status == IPX_OK) {                                                       \
            const hipError_t e_ = (expr);                                                \
            if (e_ != hipSuccess) { ipx_set_error("%s failed: %s (group rank %d)", #expr, hipGetErrorString(e_), r); rank_fail(c, rk, IPX_ERR_HIP); } \
        }                                                                                \
    } while (0)
#define RK_NCCL(expr)                                                                    \
    do {                                                                                 \
        if (rk.status == IPX_OK) {                                                       \
            const int e_ = (expr);                                                       \
            if (e_ != 0) { ipx_set_error("%s failed: %s (group rank %d)", #expr, g->rccl.GetErrorString(e_), r); rank_fail(c, rk, IPX_ERR_HIP); } \
        }                                                                                \
    } while (0)
// every rank passes the same barriers; after one, all ranks see the same `failed` and leave together
#define RK_SYNC() do { if (g->barrier.wait(c.failed)) return; } while (0)

void rank_body(Call &c, int r)
{
    ipx_group *g = c.g;
    Rank &rk = g->ranks[r];
    ipx_ctx *ctx = rk.ctx;
    const int nd = g->nd, n = c.n, H = c.H, W = c.W;
    const bool rccl = g->transport == IPX_GROUP_RCCL;
    const bool exchange = nd > 1 || rccl;                  // a world of one still runs its collectives over RCCL
    const size_t P = (size_t)H * W, words = (size_t)H * ipx_wpr(W), plane = words * 8;
    const int radius = c.p.sliding_window_radius > 1 ? c.p.sliding_window_radius : 1;
    const int T = c.p.consecutive_threshold, edge = T - 1 > 0 ? T - 1 : 0;
    int a, b;
    shard_range(n, r, nd, a, b);
    const int lo = a - radius > 0 ? a - radius : 0, hi = b + radius < n ? b + radius : n;
    const int hl = a - lo, nown = b - a, next = hi - lo;
    rk.nown = nown;
    rk.status = IPX_OK;
    rk.error[0] = 0;

    // ---- buffers (context slots: they only grow) and the upload of this rank's frames -----------------------
    void *misc = nullptr, *stack = nullptr, *bright = nullptr, *pers = nullptr, *stat = nullptr, *fin = nullptr,
         *maps = nullptr, *cnt = nullptr, *edges = nullptr, *pers_all = nullptr, *maps_all = nullptr;
    RK_HIP(hipSetDevice(rk.device));
    RK_TRY(ipx_set_params(ctx, &c.p));
    const size_t misc_bytes = ipx_align_up((size_t)next, 256) + ipx_align_up((size_t)nown, 256) +
                              ipx_align_up(sizeof(float) * (size_t)nown, 256) + sizeof(int64_t) * 8 * (size_t)nown;
    RK_TRY(ipx_scratch(ctx, GS_MISC, misc_bytes, &misc));
    RK_TRY(ipx_scratch(ctx, WS_RAW, plane * (size_t)(nown + 2 * edge), &stack));
    RK_TRY(ipx_scratch(ctx, WS_BRIGHT, plane * (size_t)nown, &bright));
    RK_TRY(ipx_scratch(ctx, WS_PERS, plane * 2, &pers));                   // [own map | merged map]
    RK_TRY(ipx_scratch(ctx, WS_STATIC, plane, &stat));
    RK_TRY(ipx_scratch(ctx, WS_FINAL, plane * (size_t)nown, &fin));
    RK_TRY(ipx_scratch(ctx, WS_MAPS, P * 4 * 2 * 2, &maps));               // [own heat | valid][summed heat | valid]
    RK_TRY(ipx_scratch(ctx, WS_COUNTS, sizeof(int64_t) * 5 * (size_t)nown, &cnt));
    if (exchange) {
        if (rccl && edge > 0) RK_TRY(ipx_scratch(ctx, GS_EDGES, plane * 2 * edge * (size_t)(nd + 1), &edges));   // [mine][all]
        RK_TRY(ipx_scratch(ctx, GS_PERS_ALL, plane * (size_t)nd, &pers_all));
        if (!rccl) RK_TRY(ipx_scratch(ctx, GS_MAPS_ALL, P * 8 * (size_t)nd, &maps_all));
    }
    char *mp = (char *)misc;
    const double *d_tab = rk.tab[c.buf];                  // uploaded by ipx_group_submit with the frames
    uint8_t *keep_ext = (uint8_t *)mp;               mp += ipx_align_up((size_t)next, 256);
    uint8_t *d_keep = (uint8_t *)mp;                 mp += ipx_align_up((size_t)nown, 256);
    float *d_flows = (float *)mp;                    mp += ipx_align_up(sizeof(float) * (size_t)nown, 256);
    int64_t *d_counts5 = (int64_t *)mp, *d_cluster = d_counts5 + 5 * (size_t)nown;
    int64_t *dense2 = (int64_t *)cnt, *fin3 = dense2 + 2 * (size_t)nown;
    uint8_t *ext_frames = rk.ext[c.buf];                  // uploaded by ipx_group_submit on the side stream
    const uint8_t *own = ext_frames + (size_t)hl * P;
    uint64_t *raw = (uint64_t *)stack + (size_t)edge * words;
    uint64_t *pers_own = (uint64_t *)pers, *pers_merged = pers_own + words;
    int32_t *maps_own = (int32_t *)maps, *maps_sum = maps_own + 2 * P;
    rk.raw = raw; rk.pers = pers_own; rk.maps = maps_own;
    hipStream_t st = rk.stream;
    if (rk.status == IPX_OK && !ext_frames) { ipx_set_error("group rank %d: no frame buffer", r); rank_fail(c, rk, IPX_ERR_HIP); }
    // (No host-to-device copy on this stream: queued behind the 36 ms uploads of two chunks on the same DMA engine, a copy of
    // a few hundred bytes held the first chunk's kernels up for as long -- measured, bench.py --group.)
    RK_HIP(hipStreamWaitEvent(st, rk.ev_up[c.buf], 0));
    RK_SYNC();

    // ---- motion test, raw planes, bright-region rejection of the owned frames ---------------------------------
    RK_TRY(ipx_flow_metric(ctx, own, nown, H, W, a > 0 ? own - P : nullptr, d_flows, d_keep));
    RK_HIP(hipMemsetAsync(keep_ext, 1, (size_t)next, st));                 // halo frames count as kept neighbours
    RK_HIP(hipMemcpyAsync(keep_ext + hl, d_keep, (size_t)nown, hipMemcpyDeviceToDevice, st));
    RK_TRY(ipx_detect_raw(ctx, ext_frames, next, H, W, hl, nown, keep_ext, raw, (uint64_t *)bright, dense2));
    RK_TRY(ipx_remove_bright_regions(ctx, (const uint64_t *)bright, raw, nown, H, W, raw));
    RK_HIP(hipEventRecord(rk.ev_raw, st));
    RK_SYNC();

    // ---- persistence across shard boundaries -------------------------------------------------------------------
    if (exchange && edge > 0) {
        uint64_t *left = (uint64_t *)stack, *right = raw + (size_t)nown * words;
        if (rccl) {
            uint64_t *mine = (uint64_t *)edges, *all = mine + 2 * (size_t)edge * words;
            RK_HIP(hipMemcpyAsync(mine, raw, plane * edge, hipMemcpyDeviceToDevice, st));
            RK_HIP(hipMemcpyAsync(mine + (size_t)edge * words, raw + (size_t)(nown - edge) * words, plane * edge, hipMemcpyDeviceToDevice, st));
            RK_SYNC();                                     // nothing fallible between this agreement and the collective
            RK_NCCL(g->rccl.AllGather(mine, all, 2 * (size_t)edge * words, NCCL_UINT64, rk.comm, st));
            if (r > 0) RK_HIP(hipMemcpyAsync(left, all + ((size_t)(r - 1) * 2 + 1) * edge * words, plane * edge, hipMemcpyDeviceToDevice, st));
            else RK_HIP(hipMemsetAsync(left, 0, plane * edge, st));
            if (r + 1 < nd) RK_HIP(hipMemcpyAsync(right, all + (size_t)(r + 1) * 2 * edge * words, plane * edge, hipMemcpyDeviceToDevice, st));
            else RK_HIP(hipMemsetAsync(right, 0, plane * edge, st));
        } else {
            if (r > 0) {
                const Rank &nb = g->ranks[r - 1];
                RK_HIP(hipStreamWaitEvent(st, nb.ev_raw, 0));
                RK_HIP(hipMemcpyPeerAsync(left, rk.device, nb.raw + (size_t)(nb.nown - edge) * words, nb.device, plane * edge, st));
            } else RK_HIP(hipMemsetAsync(left, 0, plane * edge, st));
            if (r + 1 < nd) {
                const Rank &nb = g->ranks[r + 1];
                RK_HIP(hipStreamWaitEvent(st, nb.ev_raw, 0));
                RK_HIP(hipMemcpyPeerAsync(right, rk.device, nb.raw, nb.device, plane * edge, st));
            } else RK_HIP(hipMemsetAsync(right, 0, plane * edge, st));
        }
        // (no keep list: planes of discarded frames are all zero, which resets the runs just the same)
        RK_TRY(ipx_persistent_mask(ctx, (const uint64_t *)stack, nullptr, nown + 2 * edge, H, W, pers_own));
    } else {
        RK_TRY(ipx_persistent_mask(ctx, raw, d_keep, nown, H, W, pers_own));
    }
    RK_HIP(hipEventRecord(rk.ev_pers, st));
    RK_SYNC();
    const uint64_t *pers_use = pers_own;
    if (exchange) {
        if (rccl) {
            RK_NCCL(g->rccl.AllGather(pers_own, pers_all, words, NCCL_UINT64, rk.comm, st));
        } else {
            for (int q = 0; q < nd; ++q) {
                const Rank &o = g->ranks[q];
                if (q != r) RK_HIP(hipStreamWaitEvent(st, o.ev_pers, 0));
                RK_HIP(hipMemcpyPeerAsync((uint64_t *)pers_all + (size_t)q * words, rk.device, o.pers, o.device, plane, st));
            }
        }
        RK_TRY(ipx_or_planes(ctx, (const uint64_t *)pers_all, nd, H, W, pers_merged));
        pers_use = pers_merged;
    }

    // ---- static map from chunk-wide heat / valid sums -----------------------------------------------------------
    RK_TRY(ipx_static_maps(ctx, own, raw, pers_use, d_keep, nown, H, W, maps_own, maps_own + P));
    RK_HIP(hipEventRecord(rk.ev_maps, st));
    RK_SYNC();
    const int32_t *maps_use = maps_own;
    if (exchange) {
        if (rccl) {
            RK_NCCL(g->rccl.AllReduce(maps_own, maps_sum, 2 * P, NCCL_INT32, NCCL_SUM, rk.comm, st));
        } else {
            for (int q = 0; q < nd; ++q) {
                const Rank &o = g->ranks[q];
                if (q != r) RK_HIP(hipStreamWaitEvent(st, o.ev_maps, 0));
                RK_HIP(hipMemcpyPeerAsync((int32_t *)maps_all + (size_t)q * 2 * P, rk.device, o.maps, o.device, P * 8, st));
            }
            if (rk.status == IPX_OK) {
                hipLaunchKernelGGL(sum_i32_kernel, dim3((unsigned)((2 * P + 255) / 256)), dim3(256), 0, st, (const int32_t *)maps_all, nd, 2 * P, maps_sum);
                RK_HIP(hipGetLastError());
            }
        }
        maps_use = maps_sum;
    }
    RK_TRY(ipx_static_from_maps(ctx, maps_use, maps_use + P, H, W, (uint64_t *)stat));

    // ---- final masks, counters, cluster statistics; results of the owned frames -> the caller's host arrays ------
    RK_TRY(ipx_finalize(ctx, own, raw, pers_use, (const uint64_t *)stat, d_keep, nown, H, W, (uint64_t *)fin, fin3));
    RK_TRY(ipx_cluster_stats(ctx, own, (const uint64_t *)fin, d_keep, nown, H, W, d_tab, c.tdim, d_cluster, nullptr));
    RK_TRY(ipx_merge_counts(ctx, fin3, dense2, nown, d_counts5));
    // The copy-back is enqueued only when the kernels are through: a device-to-host copy that waits in a DMA queue for the
    // chunk's kernels holds up the NEXT chunk's upload (ipx_group_submit, side stream) queued on the same engine behind it.
    RK_TRY(ipx_synchronize(ctx));                          // also reports a failure inside the flow stage's kernels
    RK_HIP(hipMemcpyAsync(c.flows + a, d_flows, sizeof(float) * (size_t)nown, hipMemcpyDeviceToHost, st));
    RK_HIP(hipMemcpyAsync(c.keep + a, d_keep, (size_t)nown, hipMemcpyDeviceToHost, st));
    RK_HIP(hipMemcpyAsync(c.counts + 5 * (size_t)a, d_counts5, sizeof(int64_t) * 5 * (size_t)nown, hipMemcpyDeviceToHost, st));
    RK_HIP(hipMemcpyAsync(c.cluster + 3 * (size_t)a, d_cluster, sizeof(int64_t) * 3 * (size_t)nown, hipMemcpyDeviceToHost, st));
    RK_HIP(hipStreamSynchronize(st));
    (void)g->barrier.wait(c.failed);                       // nobody reuses a buffer a neighbour may still be reading
}

void rank_thread(Call *c, int r)
{
    ipx_group *g = c->g;
    {   // chunks are taken strictly in order: wait until every rank is through with the chunk before
        std::unique_lock<std::mutex> lk(g->m);
        g->cv.wait(lk, [&] { return g->completed == c->seq; });
    }
    rank_body(*c, r);
    Rank &rk = g->ranks[r];
    (void)hipSetDevice(rk.device);
    if (c->failed.load() && g->transport == IPX_GROUP_RCCL) {
        // A rank that failed may have left a collective the others had already enqueued: draining the stream would wait
        // for it for ever.  Abandon the communicator instead; the group refuses further chunks.
        if (g->rccl.CommAbort && rk.comm) { (void)g->rccl.CommAbort(rk.comm); rk.comm = nullptr; }
        std::lock_guard<std::mutex> lk(g->m);
        g->broken = true;
    }
    // a rank that left early (failure seen after a barrier) still drains its stream before the chunk counts as done
    (void)hipStreamSynchronize(rk.stream);
    c->status[r] = rk.status;
    c->error[r] = rk.error;
    if (--c->ranks_left == 0) {
        std::lock_guard<std::mutex> lk(g->m);
        ++g->completed;
        g->cv.notify_all();
    }
}

}  // namespace

extern "C" int ipx_group_create(const int *devices, int ndev, int transport, ipx_group **out)
{
    IPX_REQUIRE(out, "null pointer");
    IPX_REQUIRE(ndev >= 1 && ndev <= 64, "1 to 64 ranks");
    IPX_REQUIRE(transport == IPX_GROUP_AUTO || transport == IPX_GROUP_PEER || transport == IPX_GROUP_RCCL, "unknown transport");
    int count = 0;
    IPX_CHECK_HIP(hipGetDeviceCount(&count));
    std::vector<int> dev(ndev);
    bool distinct = true;
    for (int i = 0; i < ndev; ++i) {
        dev[i] = devices ? devices[i] : i;
        IPX_REQUIRE(dev[i] >= 0 && dev[i] < count, "no such device");
        for (int j = 0; j < i; ++j) distinct = distinct && dev[j] != dev[i];
    }
    ipx_group *g = new ipx_group();
    g->nd = ndev;
    ipx_default_params(&g->p);
    if (transport != IPX_GROUP_PEER) {
        const bool loaded = distinct && (ndev > 1 || transport == IPX_GROUP_RCCL) && rccl_load(g->rccl);
        if (!loaded && transport == IPX_GROUP_RCCL) {
            ipx_set_error(distinct ? "ipx_group_create: librccl could not be loaded (%s)" : "ipx_group_create: RCCL needs distinct devices%s",
                          distinct ? dlerror() : "");
            delete g;
            return IPX_ERR_UNSUPPORTED;
        }
        transport = loaded ? IPX_GROUP_RCCL : IPX_GROUP_PEER;
    }
    g->transport = transport;
    g->ranks.resize(ndev);
    g->barrier.count = ndev;
    int status = IPX_OK;
    for (int i = 0; i < ndev && status == IPX_OK; ++i) {
        Rank &rk = g->ranks[i];
        rk.device = dev[i];
        status = ipx_ctx_create(dev[i], &rk.ctx);
        if (status != IPX_OK) break;
        hipError_t e = hipSetDevice(dev[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&rk.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rk.ev_raw, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rk.ev_pers, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rk.ev_maps, hipEventDisableTiming);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&rk.up_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rk.ev_up[0], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&rk.ev_up[1], hipEventDisableTiming);
        if (e != hipSuccess) { ipx_set_error("ipx_group_create: %s", hipGetErrorString(e)); status = IPX_ERR_HIP; break; }
        (void)ipx_set_stream(rk.ctx, rk.stream);
        if (transport == IPX_GROUP_PEER)
            for (int j = 0; j < ndev; ++j)
                if (dev[j] != dev[i]) {                    // direct xGMI copies; without it the copies are staged
                    const hipError_t pe = hipDeviceEnablePeerAccess(dev[j], 0);
                    if (pe != hipSuccess) (void)hipGetLastError();
                }
    }
    if (status == IPX_OK && transport == IPX_GROUP_RCCL) {
        std::vector<ncclComm_t> comms(ndev);
        const int e = g->rccl.CommInitAll(comms.data(), ndev, dev.data());
        if (e != 0) { ipx_set_error("ncclCommInitAll failed: %s", g->rccl.GetErrorString(e)); status = IPX_ERR_HIP; }
        else for (int i = 0; i < ndev; ++i) g->ranks[i].comm = comms[i];
    }
    if (status != IPX_OK) { (void)ipx_group_destroy(g); return status; }
    *out = g;
    return IPX_OK;
}

extern "C" int ipx_group_destroy(ipx_group *g)
{
    if (!g) return IPX_OK;
    while (!g->inflight.empty()) (void)ipx_group_wait(g);  // chunks the caller never waited for
    for (Rank &rk : g->ranks) {
        (void)hipSetDevice(rk.device);
        if (rk.stream) (void)hipStreamSynchronize(rk.stream);
        if (rk.up_stream) (void)hipStreamSynchronize(rk.up_stream);
        if (rk.comm) (void)g->rccl.CommDestroy(rk.comm);
    }
    for (Rank &rk : g->ranks) {
        (void)hipSetDevice(rk.device);
        if (rk.ctx) { (void)ipx_set_stream(rk.ctx, rk.stream); (void)ipx_ctx_destroy(rk.ctx); }
        if (rk.ev_raw) (void)hipEventDestroy(rk.ev_raw);
        if (rk.ev_pers) (void)hipEventDestroy(rk.ev_pers);
        if (rk.ev_maps) (void)hipEventDestroy(rk.ev_maps);
        if (rk.ev_up[0]) (void)hipEventDestroy(rk.ev_up[0]);
        if (rk.ev_up[1]) (void)hipEventDestroy(rk.ev_up[1]);
        if (rk.up_stream) (void)hipStreamDestroy(rk.up_stream);
        if (rk.stream) (void)hipStreamDestroy(rk.stream);
    }
    // (librccl stays loaded: unloading a library with live GPU state is not worth the risk)
    delete g;
    return IPX_OK;
}

extern "C" int ipx_group_set_params(ipx_group *g, const ipx_params *p)
{
    IPX_REQUIRE(g && p, "null pointer");
    g->p = *p;                                             // taken by the chunks submitted from now on
    return IPX_OK;
}

extern "C" int ipx_group_info(const ipx_group *g, int *ndev, int *transport)
{
    IPX_REQUIRE(g, "null pointer");
    if (ndev) *ndev = g->nd;
    if (transport) *transport = g->transport;
    return IPX_OK;
}

extern "C" int ipx_group_submit(ipx_group *g, const uint8_t *frames_host, int n, int H, int W,
                                const double *pow2_table_host, int tdim,
                                float *flows_host, uint8_t *keep_host, int64_t *counts_host, int64_t *cluster_host)
{
    IPX_REQUIRE(g && frames_host && pow2_table_host && flows_host && keep_host && counts_host && cluster_host, "null pointer");
    IPX_REQUIRE(n >= 2 && H >= 2 && W >= 2 && tdim >= 1, "bad sizes (at least 2 frames of at least 2x2)");
    IPX_REQUIRE(n / g->nd + 1 + 2 * (g->p.sliding_window_radius > 1 ? g->p.sliding_window_radius : 1) <= 65535, "at most 65535 frames (halo included) per rank");
    IPX_REQUIRE(g->p.sliding_window_radius >= 1, "sliding_window_radius must be >= 1");
    IPX_REQUIRE(g->inflight.size() < 2, "two chunks are in flight already: ipx_group_wait first");
    bool broken;
    { std::lock_guard<std::mutex> lk(g->m); broken = g->broken; }     // (written by the ranks' threads)
    if (broken) {
        ipx_set_error("ipx_group_submit: an earlier chunk abandoned an RCCL collective; the group cannot be used any more");
        return IPX_ERR_HIP;
    }
    const int edge = g->p.consecutive_threshold - 1 > 0 ? g->p.consecutive_threshold - 1 : 0;
    const int least = edge > 1 ? edge : 1;
    if (n / g->nd < least) {
        ipx_set_error("ipx_group_detect_chunk: %d frames over %d ranks: every rank needs at least max(consecutive_threshold - 1, 1) = %d frames",
                      n, g->nd, least);
        return IPX_ERR_ARG;
    }
    // ---- the upload: every rank's frame range + halo into the frame buffer the chunk before does not use, on the rank's
    // side stream, now -- under whatever the ranks are computing.  (The chunk that used this buffer last has been waited
    // for: at most two are in flight.)  Pageable host memory makes these copies synchronous: ipx_host_register it.
    const int buf = (int)(g->submitted & 1);
    const size_t P = (size_t)H * W;
    const int radius = g->p.sliding_window_radius > 1 ? g->p.sliding_window_radius : 1;
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    int status = IPX_OK;
    Call *c = new Call();
    c->tab.assign(pow2_table_host, pow2_table_host + (size_t)tdim * tdim); c->tdim = tdim;   // (outlives its upload: freed by wait)
    for (int r = 0; r < g->nd && status == IPX_OK; ++r) {
        Rank &rk = g->ranks[r];
        int a, b;
        shard_range(n, r, g->nd, a, b);
        const int lo = a - radius > 0 ? a - radius : 0, hi = b + radius < n ? b + radius : n;
        void *ext = nullptr;
        hipError_t e = hipSetDevice(rk.device);
        if (e == hipSuccess) {
            const size_t frame_bytes = ipx_align_up((size_t)(hi - lo) * P, 256), tab_bytes = sizeof(double) * (size_t)tdim * tdim;
            status = ipx_scratch(rk.ctx, buf ? GS_FRAMES_B : GS_FRAMES, frame_bytes + tab_bytes, &ext);
            if (status != IPX_OK) break;
            rk.ext[buf] = (uint8_t *)ext;
            rk.tab[buf] = (double *)((uint8_t *)ext + frame_bytes);
            // the table first: the staging copy of pageable memory then does not wait for the frames ahead of it
            e = hipMemcpyAsync(rk.tab[buf], c->tab.data(), tab_bytes, hipMemcpyHostToDevice, rk.up_stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(ext, frames_host + (size_t)lo * P, (size_t)(hi - lo) * P, hipMemcpyHostToDevice, rk.up_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(rk.ev_up[buf], rk.up_stream);
        if (e != hipSuccess) { ipx_set_error("ipx_group_submit: upload for rank %d failed: %s", r, hipGetErrorString(e)); status = IPX_ERR_HIP; }
    }
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (status != IPX_OK) {
        for (Rank &rk : g->ranks) { (void)hipSetDevice(rk.device); (void)hipStreamSynchronize(rk.up_stream); }
        if (caller_device >= 0) (void)hipSetDevice(caller_device);
        delete c;
        return status;
    }
    c->g = g; c->frames = frames_host; c->n = n; c->H = H; c->W = W;
    c->flows = flows_host; c->keep = keep_host; c->counts = counts_host; c->cluster = cluster_host;
    c->p = g->p; c->buf = buf; c->seq = g->submitted++;
    c->ranks_left.store(g->nd);
    c->status.assign(g->nd, IPX_OK);
    c->error.assign(g->nd, std::string());
    g->inflight.push_back(c);
    for (int r = 0; r < g->nd; ++r) c->th.emplace_back(rank_thread, c, r);
    return IPX_OK;
}

extern "C" int ipx_group_wait(ipx_group *g)
{
    IPX_REQUIRE(g, "null pointer");
    IPX_REQUIRE(!g->inflight.empty(), "no chunk in flight");
    Call *c = g->inflight.front();
    g->inflight.pop_front();
    for (std::thread &t : c->th) t.join();
    int status = IPX_OK;
    for (int r = 0; r < g->nd && status == IPX_OK; ++r)
        if (c->status[r] != IPX_OK) {
            ipx_set_error("group rank %d (device %d): %s", r, g->ranks[r].device, c->error[r].c_str());
            status = c->status[r];
        }
    delete c;
    return status;
}

extern "C" int ipx_group_detect_chunk(ipx_group *g, const uint8_t *frames_host, int n, int H, int W,
                                      const double *pow2_table_host, int tdim,
                                      float *flows_host, uint8_t *keep_host, int64_t *counts_host, int64_t *cluster_host)
{
    IPX_REQUIRE(g, "null pointer");
    IPX_REQUIRE(g->inflight.empty(), "chunks submitted with ipx_group_submit are still in flight: ipx_group_wait first");
    IPX_TRY(ipx_group_submit(g, frames_host, n, H, W, pow2_table_host, tdim, flows_host, keep_host, counts_host, cluster_host));
    return ipx_group_wait(g);
}

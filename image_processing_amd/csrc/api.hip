// api.hip -- context, error reporting, parameters, small helper kernels of libipx.so.
#include "ipx_internal.h"
#include <chrono>
#include <dlfcn.h>
#include <stdlib.h>
#include <math.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void ipx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *ipx_last_error(void) { return g_err; }
extern "C" int ipx_version(void) { return 100; }

extern "C" int ipx_device_count(int *count)
{
    IPX_REQUIRE(count, "null pointer");
    IPX_CHECK_HIP(hipGetDeviceCount(count));
    return IPX_OK;
}

extern "C" void ipx_default_params(ipx_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->consecutive_threshold = 2;      // ip.py:168
    p->brightness_threshold = 170;     // ip.py:169
    p->flow_threshold = 2.0;           // ip.py:170
    p->static_threshold = 50;          // ip.py:171
    p->min_cluster_size = 5;           // ip.py:172
    p->max_cluster_size = 20;          // ip.py:173
    p->min_circularity = 0.1;          // ip.py:174
    p->sliding_window_radius = 3;      // ip.py:175
    p->min_valid_frames = 10;          // ip.py:589
    p->circularity_size_threshold = 10;  // ip.py:657
    p->estimate_threshold = 170;       // ip.py:773
    p->pixel_std_coeff = 1.0;          // ip.py:204
}

extern "C" int ipx_words_per_row(int W) { return ipx_wpr(W); }

ipx_derived ipx_derive(const ipx_params &p)
{
    ipx_derived d;
    // u8 frame > t  (numpy compares exactly, whatever t's Python type)  <=>  frame > floor(t)
    double f = floor(p.brightness_threshold);
    if (f < -1) f = -1;
    if (f > 255) f = 255;
    d.frame_gt = (int)f;
    d.bg_gt = (float)p.brightness_threshold;   // f32 array > weak Python scalar: compared in f32
    d.est_lt = p.estimate_threshold;
    return d;
}

extern "C" int ipx_ctx_create(int device, ipx_ctx **out)
{
    IPX_REQUIRE(out, "null pointer");
    int count = 0;
    IPX_CHECK_HIP(hipGetDeviceCount(&count));
    IPX_REQUIRE(device >= 0 && device < count, "no such device");
    IPX_CHECK_HIP(hipSetDevice(device));
    IPX_TRY(ipx_static_scratch_init(device));
    ipx_ctx *c = new ipx_ctx();
    c->device = device;
    c->stream = nullptr;
    for (int i = 0; i < SLOT_COUNT; ++i) { c->slot[i] = nullptr; c->slot_bytes[i] = 0; }
    c->flow_plan = nullptr;
    c->copy_stream = nullptr;
    c->copy_done = nullptr;
    c->fi_spin_limit = 1u << 27;
    c->flow_ran = 0;
    memset(&c->prof, 0, sizeof(c->prof));
    ipx_default_params(&c->p);
    hipError_t e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e != hipSuccess) {
        delete c;
        ipx_set_error("hipEventCreate failed: %s", hipGetErrorString(e));
        return IPX_ERR_HIP;
    }
    *out = c;
    return IPX_OK;
}

extern "C" int ipx_ctx_destroy(ipx_ctx *ctx)
{
    if (!ctx) return IPX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < SLOT_COUNT; ++i)
        if (ctx->slot[i]) (void)hipFree(ctx->slot[i]);
    ipx_flow_plan_free(ctx->flow_plan);
    for (int i = 0; i < ctx->prof.cap; ++i) {
        (void)hipEventDestroy(ctx->prof.recs[i].e0);
        (void)hipEventDestroy(ctx->prof.recs[i].e1);
    }
    free(ctx->prof.recs);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
        (void)hipEventDestroy(ctx->copy_done);
    }
    delete ctx;
    return IPX_OK;
}

int ipx_ctx_enter(ipx_ctx *ctx)
{
    IPX_CHECK_HIP(hipSetDevice(ctx->device));
    return IPX_OK;
}

int ipx_scratch(ipx_ctx *ctx, int s, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 256;
    if (bytes > ctx->slot_bytes[s]) {
        static const bool trace = getenv("IPX_SCRATCH_TRACE") != nullptr;      // (debug aid: what a first call waits for)
        const auto t0 = std::chrono::steady_clock::now();
        // growing: everything enqueued so far may still use the old arena
        IPX_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->slot[s]) IPX_CHECK_HIP(hipFree(ctx->slot[s]));
        ctx->slot[s] = nullptr;
        ctx->slot_bytes[s] = 0;
        const size_t want = ipx_align_up(bytes + bytes / 16, 1 << 20);
        hipError_t e = hipMalloc(&ctx->slot[s], want);
        if (e != hipSuccess) {
            ipx_set_error("scratch allocation of %zu bytes (slot %d) failed: %s", want, s, hipGetErrorString(e));
            return IPX_ERR_NOMEM;
        }
        ctx->slot_bytes[s] = want;
        if (trace)
            fprintf(stderr, "[ipx scratch] slot %d: %.1f MB in %.2f ms\n", s, want / 1048576.0,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    *out = ctx->slot[s];
    return IPX_OK;
}

extern "C" int ipx_set_params(ipx_ctx *ctx, const ipx_params *p)
{
    IPX_REQUIRE(ctx && p, "null pointer");
    ctx->p = *p;
    return IPX_OK;
}

extern "C" int ipx_set_stream(ipx_ctx *ctx, void *hip_stream)
{
    IPX_REQUIRE(ctx, "null pointer");
    ctx->stream = (hipStream_t)hip_stream;
    return IPX_OK;
}

extern "C" int ipx_synchronize(ipx_ctx *ctx)
{
    IPX_REQUIRE(ctx, "null pointer");
    if (int e = ipx_ctx_enter(ctx)) return e;
    IPX_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    int raised = 0;
    IPX_TRY(ipx_flow_status(ctx, &raised, 0));
    if (raised) {
        ipx_set_error("flow stage: a hand-off inside flow_iter_kernel timed out; its metrics were reported as NaN "
                      "(frames discarded).  ipx_flow_status(ctx, &r, 1) clears the condition");
        return IPX_ERR_HIP;
    }
    return IPX_OK;
}

// ---- input side: host frames -> HBM on a side stream ---------------------------------------------------------
extern "C" int ipx_host_register(void *host, size_t bytes)
{
    IPX_REQUIRE(host && bytes > 0, "bad buffer");
    const hipError_t e = hipHostRegister(host, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();                         // not sticky: the caller may go on with pageable memory
        ipx_set_error("hipHostRegister of %zu bytes failed: %s (the copy still works from pageable memory)", bytes,
                      hipGetErrorString(e));
        return IPX_ERR_HIP;
    }
    return IPX_OK;
}

extern "C" int ipx_host_unregister(void *host)
{
    IPX_REQUIRE(host, "null pointer");
    IPX_CHECK_HIP(hipHostUnregister(host));
    return IPX_OK;
}

extern "C" int ipx_upload_frames(ipx_ctx *ctx, const uint8_t *host_frames, size_t bytes, uint8_t *dev_frames)
{
    IPX_REQUIRE(ctx && host_frames && dev_frames, "null pointer");
    if (bytes == 0) return IPX_OK;
    if (int e = ipx_ctx_enter(ctx)) return e;
    if (!ctx->copy_stream) {
        IPX_CHECK_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        IPX_CHECK_HIP(hipEventCreateWithFlags(&ctx->copy_done, hipEventDisableTiming));
    }
    IPX_CHECK_HIP(hipMemcpyAsync(dev_frames, host_frames, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    IPX_CHECK_HIP(hipEventRecord(ctx->copy_done, ctx->copy_stream));
    IPX_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->copy_done, 0));      // later work of the context sees the frames
    return IPX_OK;
}

extern "C" int ipx_timer_begin(ipx_ctx *ctx)
{
    IPX_REQUIRE(ctx, "null pointer");
    if (int e = ipx_ctx_enter(ctx)) return e;
    IPX_CHECK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    return IPX_OK;
}

extern "C" int ipx_timer_end(ipx_ctx *ctx, float *elapsed_ms)
{
    IPX_REQUIRE(ctx && elapsed_ms, "null pointer");
    if (int e = ipx_ctx_enter(ctx)) return e;
    IPX_CHECK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    IPX_CHECK_HIP(hipEventSynchronize(ctx->ev1));
    IPX_CHECK_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return IPX_OK;
}

// ---- roctx ranges ---------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)(void) = nullptr;
    Roctx()
    {
        const char *env = getenv("IPX_ROCTX");
        if (env && env[0] == '0') return;
        // (rocprofv3 listens to the rocprofiler-sdk flavour of roctx; the roctracer one is the fallback)
        static const char *const names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                                            "libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so"};
        for (const char *n : names) {
            void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)(void))dlsym(h, "roctxRangePop");
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const Roctx &roctx() { static const Roctx r; return r; }     // thread-safe initialisation (C++11)
}  // namespace

void ipx_range_push(const char *name) { if (roctx().push) (void)roctx().push(name); }
void ipx_range_pop(void) { if (roctx().pop) (void)roctx().pop(); }

// ---- built-in profiler ----------------------------------------------------------------------
static const char *const g_kid_names[KID_COUNT] = {
    "pyr_level", "polyexp", "flow_iter",
    "detect", "damaged_mask",
    "ccl_init", "ccl_merge", "ccl_reduce", "ccl_select", "close_cross3", "cluster_eval",
    "persistent", "static_maps", "static_from_maps", "finalize", "misc",
};

static int prof_fold(ipx_ctx *ctx)
{
    ipx_prof &pr = ctx->prof;
    if (pr.nrecs == 0) return IPX_OK;
    for (int i = 0; i < pr.nrecs; ++i) {
        float ms = 0.f;
        IPX_CHECK_HIP(hipEventSynchronize(pr.recs[i].e1));      // waits for this launch only
        IPX_CHECK_HIP(hipEventElapsedTime(&ms, pr.recs[i].e0, pr.recs[i].e1));
        pr.ms[pr.recs[i].kid] += ms;
        pr.bytes[pr.recs[i].kid] += pr.recs[i].bytes;
        pr.launches[pr.recs[i].kid] += 1;
        const int lv = pr.recs[i].level;
        if (lv >= 0 && lv < IPX_PROF_LEVELS) { pr.lvl_ms[lv] += ms; pr.lvl_bytes[lv] += pr.recs[i].bytes; pr.lvl_launches[lv] += 1; }
    }
    pr.nrecs = 0;
    return IPX_OK;
}

void ipx_prof_begin(ipx_ctx *ctx, int kid, double bytes, int level)
{
    ipx_prof &pr = ctx->prof;
    if (pr.nrecs == pr.cap) {
        if (pr.cap >= (1 << 15)) {                 // bound the pool: fold what is pending (synchronises)
            if (prof_fold(ctx) != IPX_OK) { pr.on = 0; return; }
        } else {
            const int ncap = pr.cap ? pr.cap * 2 : 256;
            ipx_prof_rec *nr = (ipx_prof_rec *)realloc(pr.recs, sizeof(ipx_prof_rec) * ncap);
            if (!nr) { pr.on = 0; return; }
            pr.recs = nr;
            for (int i = pr.cap; i < ncap; ++i) {
                if (hipEventCreate(&nr[i].e0) != hipSuccess || hipEventCreate(&nr[i].e1) != hipSuccess) { pr.on = 0; pr.cap = i; return; }
            }
            pr.cap = ncap;
        }
    }
    ipx_prof_rec &r = pr.recs[pr.nrecs];
    r.kid = kid; r.bytes = bytes; r.level = level;
    (void)hipEventRecord(r.e0, ctx->stream);
}

void ipx_prof_end(ipx_ctx *ctx)
{
    ipx_prof &pr = ctx->prof;
    if (!pr.on) return;
    (void)hipEventRecord(pr.recs[pr.nrecs].e1, ctx->stream);
    pr.nrecs++;
}

extern "C" int ipx_profile_enable(ipx_ctx *ctx, int on)
{
    IPX_REQUIRE(ctx, "null pointer");
    if (int e = ipx_ctx_enter(ctx)) return e;
    if (int e = prof_fold(ctx)) return e;
    ctx->prof.on = on ? 1 : 0;
    return IPX_OK;
}

extern "C" int ipx_profile_reset(ipx_ctx *ctx)
{
    IPX_REQUIRE(ctx, "null pointer");
    if (int e = ipx_ctx_enter(ctx)) return e;
    if (int e = prof_fold(ctx)) return e;
    for (int k = 0; k < KID_COUNT; ++k) { ctx->prof.ms[k] = 0; ctx->prof.bytes[k] = 0; ctx->prof.launches[k] = 0; }
    for (int k = 0; k < IPX_PROF_LEVELS; ++k) { ctx->prof.lvl_ms[k] = 0; ctx->prof.lvl_bytes[k] = 0; ctx->prof.lvl_launches[k] = 0; }
    return IPX_OK;
}

extern "C" int ipx_profile_kernel_classes(void) { return KID_COUNT; }

extern "C" int ipx_profile_read(ipx_ctx *ctx, int kid, const char **name, uint64_t *launches, double *total_ms,
                                double *total_bytes)
{
    IPX_REQUIRE(ctx && kid >= 0 && kid < KID_COUNT, "bad kernel class");
    if (int e = ipx_ctx_enter(ctx)) return e;
    if (int e = prof_fold(ctx)) return e;
    if (name) *name = g_kid_names[kid];
    if (launches) *launches = ctx->prof.launches[kid];
    if (total_ms) *total_ms = ctx->prof.ms[kid];
    if (total_bytes) *total_bytes = ctx->prof.bytes[kid];
    return IPX_OK;
}

extern "C" int ipx_profile_read_flow_level(ipx_ctx *ctx, int level, uint64_t *launches, double *total_ms, double *total_bytes)
{
    IPX_REQUIRE(ctx && level >= 0 && level < IPX_PROF_LEVELS, "bad pyramid level");
    if (int e = ipx_ctx_enter(ctx)) return e;
    if (int e = prof_fold(ctx)) return e;
    if (launches) *launches = ctx->prof.lvl_launches[level];
    if (total_ms) *total_ms = ctx->prof.lvl_ms[level];
    if (total_bytes) *total_bytes = ctx->prof.lvl_bytes[level];
    return IPX_OK;
}

// ---- helper kernels ------------------------------------------------------------------------
namespace {

// cv2.cvtColor(BGR2GRAY) on u8, OpenCV 4.x 15-bit fixed point, ip.py:79.
__global__ void bgr2gray_kernel(const uint8_t *bgr, size_t npix, uint8_t *gray)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < npix; i += stride) {
        const unsigned b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15);
    }
}

// one wave per 64-pixel word: byte mask -> ballot word
__global__ void pack_kernel(const uint8_t *mask, int rows, int W, int wpr, uint64_t *bits)
{
    const int lane = threadIdx.x & 63;
    const size_t word = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (word >= (size_t)rows * wpr) return;
    const size_t row = word / wpr;
    const int k = (int)(word - row * wpr);
    const int x = k * 64 + lane;
    const bool on = x < W && mask[row * W + x] != 0;
    const unsigned long long b = __ballot(on);
    if (lane == 0) bits[word] = b;
}

__global__ void unpack_kernel(const uint64_t *bits, int rows, int W, int wpr, uint8_t *mask)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * W) return;
    const size_t row = i / W;
    const int x = (int)(i - row * W);
    mask[i] = (bits[row * wpr + (x >> 6)] >> (x & 63)) & 1;
}

// Frames as an uncompressed AVI stores them (DIBs: rows of `stride` bytes, padded to 4, usually bottom-up; frame i at
// raw + i * rec) -> the (n,H,W) top-down uint8 frames cv2.VideoCapture + cvtColor hand to the path (ip.py:62-85).
// 8-bit streams (grey ramp palette): a copy with the rows turned over.  VEC = 4: four pixels per thread as one dword
// (W % 4 == 0 and 4-byte aligned records), else byte by byte.
template <int VEC>
__global__ void dib8_unpack_kernel(const uint8_t *raw, size_t rec, int n, int H, int W, int stride, int bottom_up, uint8_t *frames)
{
    const size_t per_row = (size_t)(W / VEC);
    const size_t total = (size_t)n * H * per_row;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        const size_t row = i / per_row;
        const int x = (int)(i - row * per_row) * VEC;
        const size_t f = row / H;
        const int y = (int)(row - f * H);
        const uint8_t *src = raw + f * rec + (size_t)(bottom_up ? H - 1 - y : y) * stride + x;
        uint8_t *dst = frames + row * W + x;
        if (VEC == 4) *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(src);
        else *dst = *src;
    }
}

// 24-bit streams: the same walk with OpenCV's fixed-point luma (bgr2gray_kernel) applied on the way.
__global__ void dib24_gray_kernel(const uint8_t *raw, size_t rec, int n, int H, int W, int stride, int bottom_up, uint8_t *frames)
{
    const size_t total = (size_t)n * H * W;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < total; i += step) {
        const size_t row = i / W;
        const int x = (int)(i - row * W);
        const size_t f = row / H;
        const int y = (int)(row - f * H);
        const uint8_t *src = raw + f * rec + (size_t)(bottom_up ? H - 1 - y : y) * stride + 3 * (size_t)x;
        const unsigned b = src[0], g = src[1], r = src[2];
        frames[i] = (uint8_t)((b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15);
    }
}

}  // namespace

extern "C" int ipx_unpack_dib(ipx_ctx *ctx, const uint8_t *raw, size_t rec_bytes, int n, int H, int W, int bits_per_pixel,
                              int stride, int bottom_up, uint8_t *frames)
{
    IPX_REQUIRE(ctx && raw && frames, "null pointer");
    IPX_REQUIRE(n > 0 && H > 0 && W > 0, "bad sizes");
    IPX_REQUIRE(bits_per_pixel == 8 || bits_per_pixel == 24, "8-bit (grey ramp) or 24-bit BGR frames");
    IPX_REQUIRE(stride >= W * (bits_per_pixel / 8) && rec_bytes >= (size_t)stride * H, "rows / records shorter than the image");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const size_t P = (size_t)n * H * W;
    if (bits_per_pixel == 8) {
        const bool vec = (W & 3) == 0 && (stride & 3) == 0 && (rec_bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(raw) & 3) == 0 &&
                         (reinterpret_cast<uintptr_t>(frames) & 3) == 0;
        const size_t items = vec ? P / 4 : P;
        const unsigned blocks = (unsigned)((items + 255) / 256 < 65536 ? (items + 255) / 256 : 65536);
        if (vec) IPX_LAUNCH(ctx, KID_MISC, 2.0 * P, dib8_unpack_kernel<4>, dim3(blocks), dim3(256), raw, rec_bytes, n, H, W, stride, bottom_up, frames);
        else IPX_LAUNCH(ctx, KID_MISC, 2.0 * P, dib8_unpack_kernel<1>, dim3(blocks), dim3(256), raw, rec_bytes, n, H, W, stride, bottom_up, frames);
    } else {
        const unsigned blocks = (unsigned)((P + 255) / 256 < 65536 ? (P + 255) / 256 : 65536);
        IPX_LAUNCH(ctx, KID_MISC, 4.0 * P, dib24_gray_kernel, dim3(blocks), dim3(256), raw, rec_bytes, n, H, W, stride, bottom_up, frames);
    }
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_bgr2gray(ipx_ctx *ctx, const uint8_t *bgr, size_t npix, uint8_t *gray)
{
    IPX_REQUIRE(ctx && bgr && gray, "null pointer");
    if (npix == 0) return IPX_OK;
    if (int e = ipx_ctx_enter(ctx)) return e;
    const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    IPX_LAUNCH(ctx, KID_MISC, 4.0 * npix, bgr2gray_kernel, dim3(blocks), dim3(256), bgr, npix, gray);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_pack_mask(ipx_ctx *ctx, const uint8_t *mask, int count, int H, int W, uint64_t *bits)
{
    IPX_REQUIRE(ctx && mask && bits, "null pointer");
    IPX_REQUIRE(count > 0 && H > 0 && W > 0, "bad sizes");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const int wpr = ipx_wpr(W);
    const size_t words = (size_t)count * H * wpr;
    IPX_LAUNCH(ctx, KID_MISC, (double)count * H * W + words * 8.0, pack_kernel, dim3((unsigned)((words + 3) / 4)), dim3(256), mask,
                       count * H, W, wpr, bits);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

extern "C" int ipx_unpack_mask(ipx_ctx *ctx, const uint64_t *bits, int count, int H, int W, uint8_t *mask)
{
    IPX_REQUIRE(ctx && mask && bits, "null pointer");
    IPX_REQUIRE(count > 0 && H > 0 && W > 0, "bad sizes");
    if (int e = ipx_ctx_enter(ctx)) return e;
    const size_t n = (size_t)count * H * W;
    IPX_LAUNCH(ctx, KID_MISC, (double)n + (double)count * H * ipx_wpr(W) * 8, unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), bits,
                       count * H, W, ipx_wpr(W), mask);
    IPX_LAUNCH_CHECK();
    return IPX_OK;
}

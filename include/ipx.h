/*
 * ipx.h -- C ABI of libipx.so: the MI355X (gfx950) damaged-pixel detection path.
 *
 * The reference (Ben-Folkard/image_processing) has no FFI layer: its boundary is the
 * Python module namespace `image_processing` (image_processing.py, "ip.py" below).
 * Every entry point here names the ip.py function (file:line) whose arithmetic it
 * replaces; image_processing.py in this repository binds them with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes.  Every function returns 0 on success and a
 *    negative ipx_status on failure; ipx_last_error() returns a thread-local message.
 *  - All data pointers are DEVICE pointers on the context's GPU unless the name ends
 *    in _host.  Inputs/outputs are caller-owned (e.g. torch tensors); scratch memory
 *    is owned by the context.  Work is enqueued on the context's stream
 *    (ipx_set_stream) and is asynchronous unless stated.
 *  - frames:   uint8  [n][H][W], rows contiguous (pitch = W).
 *  - bit-plane: uint64 [H][wpr], wpr = ipx_words_per_row(W) = ceil(W/64); bit b of word k
 *    of row y is pixel (x = 64k+b, y); padding bits are zero.  A stack of planes is
 *    [n][H*wpr].
 *  - keep:     uint8 [n]; 0 marks a DISCARDED frame (the `None` masks of ip.py:557).
 */
#ifndef IPX_H
#define IPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the declarations below are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef enum ipx_status {
    IPX_OK = 0,
    IPX_ERR_ARG = -1,      /* bad argument (null pointer, non-positive size, ...) */
    IPX_ERR_HIP = -2,      /* a HIP runtime call failed */
    IPX_ERR_NOMEM = -3,    /* scratch allocation failed */
    IPX_ERR_UNSUPPORTED = -4
} ipx_status;

/* The 9 user settings of ip.py:167-177 (number_of_plots is plot-only and omitted) plus the
 * constants the reference hard-codes on this path. */
typedef struct ipx_params {
    int32_t consecutive_threshold;       /* ip.py:168, used at :289 (longest >= T)            */
    int32_t sliding_window_radius;       /* ip.py:175, used at :189-190                        */
    int32_t min_cluster_size;            /* ip.py:172, :678                                    */
    int32_t max_cluster_size;            /* ip.py:173, :678 and :753 (area >= max => remove)   */
    int32_t min_valid_frames;            /* 10, hard-coded at ip.py:589                        */
    int32_t circularity_size_threshold;  /* 10, default at ip.py:657                           */
    int32_t estimate_threshold;          /* 170, default at ip.py:773 (not forwarded at :836)  */
    int32_t reserved0;
    double brightness_threshold;         /* ip.py:169, :457 (frame > t), :743 (bg > t)         */
    double flow_threshold;               /* ip.py:170, :559 (flow <= t keeps the frame)        */
    double static_threshold;             /* ip.py:171, :465                                    */
    double min_circularity;              /* ip.py:174, :701 (compared in float32)              */
    double pixel_std_coeff;              /* 1.0, default at ip.py:204                          */
} ipx_params;

typedef struct ipx_ctx ipx_ctx;

/* ---- library / context ------------------------------------------------------------------ */
const char *ipx_last_error(void);
int ipx_version(void);
int ipx_device_count(int *count);
void ipx_default_params(ipx_params *p);                         /* ip.py:167-177 defaults    */
int ipx_words_per_row(int W);

int ipx_ctx_create(int device, ipx_ctx **out);
int ipx_ctx_destroy(ipx_ctx *ctx);
int ipx_set_params(ipx_ctx *ctx, const ipx_params *p);          /* replaces ip.py:154-180    */
int ipx_set_stream(ipx_ctx *ctx, void *hip_stream);             /* NULL = default stream     */
int ipx_synchronize(ipx_ctx *ctx);
/* Timing of the kernels enqueued between begin/end on the context's stream (HIP events). */
int ipx_timer_begin(ipx_ctx *ctx);
int ipx_timer_end(ipx_ctx *ctx, float *elapsed_ms);             /* synchronises               */

/* Built-in per-kernel profiler (no counterpart in the reference; bench.py's roofline uses it).
 * While enabled, HIP events are recorded on the context's stream around every kernel launch and
 * accumulated per kernel class with the launch's algorithmic bytes.  ipx_profile_read
 * synchronises the stream. */
int ipx_profile_enable(ipx_ctx *ctx, int on);
int ipx_profile_reset(ipx_ctx *ctx);
int ipx_profile_kernel_classes(void);
int ipx_profile_read(ipx_ctx *ctx, int kernel_class, const char **name, uint64_t *launches,
                     double *total_ms, double *total_bytes);
/* the class "flow_iter" once more, split by the pyramid level of the launch (0 = full resolution; levels 0..7):
 * the level-0 launches are 70 % of the class's time and run at another rate than the small levels */
int ipx_profile_read_flow_level(ipx_ctx *ctx, int level, uint64_t *launches, double *total_ms, double *total_bytes);

/* ---- a1 (input side): decoded frames in host memory -> HBM, ip.py:1208-1210 hands the chunk over -------------
 * The reference keeps a chunk in host RAM; here it has to reach HBM.  ipx_upload_frames enqueues the copy on
 * a side stream of the context (not on its compute stream: the upload of chunk k+1 runs under the kernels of
 * chunk k) and makes every LATER call on the context wait for it.  The copy is asynchronous to the host only
 * for pinned memory (hipHostMalloc, or any buffer passed through ipx_host_register); host_frames must stay
 * valid, and dev_frames unused by earlier work, until the copy has run (double-buffer both).
 * ipx_host_register / _unregister pin a caller-owned buffer (numpy arrays, decoder output). */
int ipx_host_register(void *host, size_t bytes);
int ipx_host_unregister(void *host);
int ipx_upload_frames(ipx_ctx *ctx, const uint8_t *host_frames, size_t bytes, uint8_t *dev_frames);

/* ---- a1: load_video_frames' grey conversion, ip.py:78-79 (cv2.cvtColor BGR2GRAY) --------- */
int ipx_bgr2gray(ipx_ctx *ctx, const uint8_t *bgr, size_t npix, uint8_t *gray);
/* The same step for frames still in the layout an uncompressed AVI stores them in (what cv2.VideoCapture.read +
 * cvtColor do on the CPU in ip.py:62-85): n device-resident DIBs, frame i at raw + i * rec_bytes, rows of `stride`
 * bytes (padded to 4), bottom-up when bottom_up != 0, 8 bits per pixel (grey-ramp palette: the indices are the grey
 * levels) or 24 (BGR, converted with the luma of ipx_bgr2gray) -> frames uint8 [n][H][W], top-down.  Lets a decoder
 * thread read a whole chunk of the file with a few large reads into pinned memory and leave the row flip, the
 * padding and the colour conversion to the GPU. */
int ipx_unpack_dib(ipx_ctx *ctx, const uint8_t *raw, size_t rec_bytes, int n, int H, int W, int bits_per_pixel,
                   int stride, int bottom_up, uint8_t *frames);

/* ---- a3: compute_optical_flow_metric, ip.py:852-873 --------------------------------------
 * flows[0] = 0 (or the metric against prev_frame when given: the left halo of a frame-range
 * shard); flows[i] = mean Farneback flow magnitude between frames i-1 and i, as float32: the f32
 * magnitudes are added in float64 in a fixed order and the mean is rounded to f32 once.  (The
 * reference's np.mean of the f32 array, ip.py:868, is numpy's f32 pairwise sum: the two agree
 * to ~1e-7 relative, inside the 1e-5 of the float metrics, but not bit for bit, so a frame whose
 * metric lies within ~1e-7 relative of flow_threshold could be kept here and discarded there;
 * bench.py --motion and tests/test_gpu_configs.py report how close frames come to it.)
 * keep[i] = (double)flows[i] <= flow_threshold (ip.py:559).
 * Frames of 2x2 up to 2^26 pixels (IPX_ERR_UNSUPPORTED beyond: 32-bit buffer offsets).
 * Failure inside the stage's kernels (a bounded wait between its waves running out) is not silent:
 * every metric of the call is then NaN with keep = 0, ipx_synchronize returns IPX_ERR_HIP and
 * ipx_flow_status reads (and clears) the condition. */
int ipx_flow_metric(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W,
                    const uint8_t *prev_frame, float *flows, uint8_t *keep);
int ipx_flow_status(ipx_ctx *ctx, int *raised, int clear);     /* synchronises the stream        */
/* No counterpart in the reference: makes the stage's plan (pyramid geometry, upsampling tables) and its two large
 * scratch arenas (tens of GB for a 1000-frame 1080p chunk) for chunks of up to n frames of H x W NOW instead of inside
 * the first ipx_flow_metric / ipx_detect_chunk call -- for a driver that has time to spare while its first chunk is
 * still being read.  Changes no result. */
int ipx_reserve_flow(ipx_ctx *ctx, int n, int H, int W);
/* test hook: number of polls after which a wave of flow_iter_kernel gives up waiting for its
 * partner waves (default 2^27, ~10 s); 0 forces the failure path described above. */
int ipx_debug_fi_spin_limit(ipx_ctx *ctx, unsigned limit);
/* One Farneback pair on float32 images, full flow field (H,W,2) out: unit parity of
 * cv2.calcOpticalFlowFarneback(prev,next,None,0.5,3,15,3,5,2.0,0), ip.py:858-867.
 * The field of a call during which the stage's error word was raised (see ipx_flow_metric) is
 * not valid: ipx_synchronize / ipx_flow_status report the condition, the field itself carries no mark. */
int ipx_farneback_pair(ipx_ctx *ctx, const float *prev, const float *next, int H, int W, float *flow);

/* ---- a4/a5: compute_background + _find_background, ip.py:183-252 -------------------------- */
int ipx_background(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W, int index, float *bg);

/* ---- a6/a7: _raw_damaged_mask / get_damaged_pixel_mask, ip.py:255-267, 635-648 ------------
 * bg_is_f64 selects the dtype the threshold chain of ip.py:638 runs in.  mask: uint8 (H,W);
 * thr: same dtype as bg (may be NULL). */
int ipx_damaged_mask(ipx_ctx *ctx, const uint8_t *frame, int H, int W, const void *bg,
                     int bg_is_f64, uint8_t *mask, void *thr);

/* ---- a4-a7 fused for a range of frames of a resident chunk (the hot kernel) ---------------
 * For frames first..first+count-1 of the n-frame chunk with keep != 0: raw damaged bit-plane
 * (ip.py:565) and bright-background bit-plane (background > brightness_threshold, ip.py:743).
 * dense_counts (may be NULL): int64 [count][2] = { #(frame < estimate_threshold),
 * #(frame > brightness_threshold) } per kept frame (inputs of a13/a14).
 * Planes of discarded frames are zeroed. */
int ipx_detect_raw(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W, int first, int count,
                   const uint8_t *keep, uint64_t *raw_bits, uint64_t *bright_bits,
                   int64_t *dense_counts);

/* ---- a8: remove_bright_regions, ip.py:731-767, on `count` independent planes ---------------
 * bright_bits: (background > threshold); damaged_bits in, cleaned bits out (may alias in). */
int ipx_remove_bright_regions(ipx_ctx *ctx, const uint64_t *bright_bits, const uint64_t *damaged_bits,
                              int count, int H, int W, uint64_t *out_bits);

/* ---- the same for callers that cannot allocate --------------------------------------------
 * No counterpart in the reference.  For callers that must not reach hipMalloc / hipFree: code under
 * HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same arguments, same
 * kernels and bit-identical results as ipx_remove_bright_regions, but the labelling scratch
 * (parent + accumulator per segment id and one `remove` plane per plane in flight) is a file-scope
 * __device__ array of IPX_NOALLOC_SCRATCH_BYTES (65 MiB, one per GPU the library runs on, resolved
 * by ipx_ctx_create) instead of the context's growable arena, and the built-in profiler is not
 * applied to its launches.  Planes of more than IPX_NOALLOC_MAX_WORDS words (H * ceil(W/64); 3840 x
 * 2160 has 129,600) are refused with IPX_ERR_UNSUPPORTED.  Every call on a GPU uses that GPU's one
 * array: calls that overlap in time on one GPU (two streams, two contexts, two threads) race;
 * calls ordered on one stream do not, and nothing has to be cleared between them. */
#define IPX_NOALLOC_MAX_WORDS 131072
#define IPX_NOALLOC_SCRATCH_BYTES ((size_t)IPX_NOALLOC_MAX_WORDS * (64 * 8 + 8))
size_t ipx_noalloc_scratch_bytes(void);
int ipx_remove_bright_regions_noalloc(ipx_ctx *ctx, const uint64_t *bright_bits, const uint64_t *damaged_bits,
                                      int count, int H, int W, uint64_t *out_bits);

/* ---- a9: compute_persistent_mask, ip.py:270-289 -------------------------------------------
 * Planes of discarded frames (keep[i] == 0, or all-zero planes when keep is NULL) reset the
 * runs, as the `None` entries do at ip.py:281-282.  A frame-range shard calls this on
 * [T-1 planes of the left neighbour | its own planes | T-1 planes of the right neighbour] and
 * ORs the shards' results (ipx_or_planes): every run of >= T frames has T consecutive frames
 * inside two adjacent shards as long as each shard holds at least T-1 frames. */
int ipx_persistent_mask(ipx_ctx *ctx, const uint64_t *raw_bits, const uint8_t *keep, int n, int H, int W,
                        uint64_t *persistent_bits);
/* out = planes[0] | ... | planes[count-1]  (merge of the shards' persistence maps; no
 * counterpart in the single-process reference).  out may alias planes[0]. */
int ipx_or_planes(ipx_ctx *ctx, const uint64_t *planes, int count, int H, int W, uint64_t *out);

/* ---- a10+a11 (first half): heat / valid maps of compute_static_mask, ip.py:451-459 ----------
 * clean = raw & ~persistent for kept frames, zero otherwise (ip.py:306, :451).
 * heat, valid: int32 (H,W), OVERWRITTEN with this call's n frames (sum across shards with
 * an all-reduce before ipx_static_from_maps). */
int ipx_static_maps(ipx_ctx *ctx, const uint8_t *frames, const uint64_t *raw_bits,
                    const uint64_t *persistent_bits, const uint8_t *keep, int n, int H, int W,
                    int32_t *heat, int32_t *valid);
/* ---- a11 (second half): percentage map > static_threshold, ip.py:461-465 ------------------- */
int ipx_static_from_maps(ipx_ctx *ctx, const int32_t *heat, const int32_t *valid, int H, int W,
                         uint64_t *static_bits);

/* ---- a10+a12+a13+a14 counters: final masks and the per-frame integer counters ---------------
 * final = raw & ~(persistent | static) for kept frames (ip.py:306, :469-478), zero planes for
 * discarded ones.  counts: int64 [n][3] = { sum(final), #(final & frame < estimate_threshold),
 * #(final & frame > brightness_threshold) } (ip.py:800-805, :840). */
int ipx_finalize(ipx_ctx *ctx, const uint8_t *frames, const uint64_t *raw_bits,
                 const uint64_t *persistent_bits, const uint64_t *static_bits, const uint8_t *keep,
                 int n, int H, int W, uint64_t *final_bits, int64_t *counts);

/* ---- a13/a14, mask-independent part: counts int64 [n][2] = { #(frame < estimate_threshold),
 * #(frame > brightness_threshold) } (ip.py:800-805, :840); ipx_detect_raw produces the same
 * numbers as a by-product for the fused path. */
int ipx_dense_counts(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W, int64_t *counts);

/* ---- a16/a17: compute_cluster_stats / filter_damaged_pixel_clusters, ip.py:332-348, 651-728 --
 * Per frame with keep != 0: morphological close (3x3 cross), 8-connected components, size
 * filter, circularity filter via the outer-contour perimeter; stats: int64 [n][3] =
 * { cluster_count, sum of kept areas, sum of frame brightness over kept components }.
 * cleaned_bits (may be NULL): the cleaned mask of ip.py:704.
 * pow2_table: host-built doubles [(tdim)*(tdim)], entry [a*tdim+d] = (a + d*sqrt2f)**2 as the
 * reference's Python evaluates it (libm pow is not always p*p). */
int ipx_cluster_stats(ipx_ctx *ctx, const uint8_t *frames, const uint64_t *mask_bits, const uint8_t *keep,
                      int n, int H, int W, const double *pow2_table, int tdim,
                      int64_t *stats, uint64_t *cleaned_bits);

/* ---- a18: detect_damaged_pixels, ip.py:526-632, whole resident chunk -----------------------
 * flows: float32 [n] out;  keep: uint8 [n] out;  counts: int64 [n][5] out =
 * { sum(final), #(final&lt), #(final&gt), #(frame<estimate_threshold), #(frame>brightness_threshold) };
 * cluster: int64 [n][3] out;  final_bits: [n][H*wpr] out (may be NULL).
 * The host turns the integer counters into the four float64 result arrays (ip.py:605-613). */
int ipx_detect_chunk(ipx_ctx *ctx, const uint8_t *frames, int n, int H, int W,
                     const double *pow2_table, int tdim,
                     float *flows, uint8_t *keep, int64_t *counts, int64_t *cluster,
                     uint64_t *final_bits);

/* ---- a18 across the GPUs of a node (BASELINE configs[2]; no counterpart in the single-process reference, whose
 * unit of work is the chunk, ip.py:1204-1227) ------------------------------------------------------------------
 * A group owns one context, one stream and one host thread per listed device.  ipx_group_detect_chunk takes the
 * chunk in HOST memory (as the reference holds it), shards it by frame range (rank r owns frames
 * [r n / ndev, (r+1) n / ndev) and uploads them plus `sliding_window_radius` halo frames each side), runs the stages
 * of ipx_detect_chunk on every shard and exchanges what crosses a shard boundary: the T-1 edge planes and the
 * persistence maps (ip.py:270-289) and the chunk-wide heat / valid maps (ip.py:451-459).  Results are identical to
 * ipx_detect_chunk on the whole chunk, bit for bit (the metric of a shard's first frame included: it is computed
 * against the left halo frame), and are written to the caller's HOST arrays: flows float32 [n], keep uint8 [n],
 * counts int64 [n][5], cluster int64 [n][3] (layouts of ipx_detect_chunk); pow2_table_host as there, but in host
 * memory.  The call returns when all ranks are done.  Every rank needs at least max(consecutive_threshold-1, 1)
 * frames.  transport: IPX_GROUP_RCCL = ncclAllGather / ncclAllReduce on an ncclCommInitAll communicator (librccl
 * is loaded at run time; devices must be distinct; a group of one still runs its collectives), IPX_GROUP_PEER =
 * hipMemcpyPeerAsync between the ranks' buffers (a device may be listed more than once), IPX_GROUP_AUTO = RCCL
 * for two or more distinct devices when librccl loads, else peer copies.  devices == NULL means 0 .. ndev-1.
 * One calling thread per group.  When a rank fails, all ranks stop at their next meeting point and the call returns
 * that rank's status and message (over RCCL the communicator is then abandoned, ncclCommAbort, and the group takes
 * no further chunks).
 * ipx_group_submit / ipx_group_wait: the same chunk call in two halves, so that two chunks can be in flight.  submit
 * starts the upload of every rank's frames at once (side stream, second frame buffer: it runs under the chunk the
 * ranks are still computing) and returns; wait returns when the OLDEST submitted chunk is complete, with its status.
 * At most two chunks in flight; the host arrays of a chunk (frames in, results out) must stay valid and untouched
 * until its wait returns; frames in pageable memory make the upload synchronous (pin them: ipx_host_register).
 * ipx_group_set_params applies to the chunks submitted after it.  ipx_group_detect_chunk = submit + wait. */
typedef struct ipx_group ipx_group;
#define IPX_GROUP_AUTO 0
#define IPX_GROUP_PEER 1
#define IPX_GROUP_RCCL 2
int ipx_group_create(const int *devices, int ndev, int transport, ipx_group **out);
int ipx_group_destroy(ipx_group *g);
int ipx_group_set_params(ipx_group *g, const ipx_params *p);
int ipx_group_info(const ipx_group *g, int *ndev, int *transport);
int ipx_group_detect_chunk(ipx_group *g, const uint8_t *frames_host, int n, int H, int W,
                           const double *pow2_table_host, int tdim,
                           float *flows_host, uint8_t *keep_host, int64_t *counts_host, int64_t *cluster_host);
int ipx_group_submit(ipx_group *g, const uint8_t *frames_host, int n, int H, int W,
                     const double *pow2_table_host, int tdim,
                     float *flows_host, uint8_t *keep_host, int64_t *counts_host, int64_t *cluster_host);
int ipx_group_wait(ipx_group *g);

/* ---- helpers: host<->device packing used by the Python stage wrappers ----------------------- */
int ipx_pack_mask(ipx_ctx *ctx, const uint8_t *mask, int count, int H, int W, uint64_t *bits);
int ipx_unpack_mask(ipx_ctx *ctx, const uint64_t *bits, int count, int H, int W, uint8_t *mask);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* IPX_H */

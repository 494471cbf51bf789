"""Engine: one libipx context on one MI355X, with torch supplying device memory and streams.

torch is plumbing here (allocation, H2D/D2H copies, the current HIP stream, torch.distributed);
every computation is a libipx call through the C ABI of include/ipx.h.  Tensors handed to the
methods must live on the engine's device; ``to_device`` uploads numpy arrays.
"""
import ctypes as C

import numpy as np
import torch

from . import _ipx
from . import hostmath

lib = _ipx.lib
check = _ipx.check


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    def __init__(self, device=0, params=None):
        if not torch.cuda.is_available():
            raise RuntimeError("image_processing_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", int(device))
        self._ctx = C.c_void_p()
        check(lib.ipx_ctx_create(int(device), C.byref(self._ctx)))
        self.params = _ipx.default_params()
        self._pow2 = None
        self._pow2_dim = 0
        self._pow2_cache = {}
        self.set_params(params)

    def close(self):
        if self._ctx:
            lib.ipx_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- settings (ip.py:154-180) ---------------------------------------------------------------
    def set_params(self, params=None):
        """``params``: dict with the reference's setting names (ip.py:167-177); missing keys keep
        the reference defaults.  Extra keys: min_valid_frames, circularity_size_threshold,
        estimate_threshold, pixel_std_coeff (constants the reference hard-codes)."""
        p = _ipx.default_params()
        d = dict(params or {})
        d.pop("number_of_plots", None)
        for k, v in d.items():
            if not hasattr(p, k):
                raise KeyError("unknown setting %r" % (k,))
            setattr(p, k, v)
        self.params = p
        check(lib.ipx_set_params(self._ctx, C.byref(p)))
        key = int(p.max_cluster_size)
        if key not in self._pow2_cache:
            tab, dim = hostmath.pow2_table(key)
            self._pow2_cache[key] = (torch.from_numpy(tab).to(self.device), dim)
        self._pow2, self._pow2_dim = self._pow2_cache[key]

    # ---- plumbing --------------------------------------------------------------------------------
    def _enter(self):
        check(lib.ipx_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def synchronize(self):
        check(lib.ipx_synchronize(self._ctx))

    def flow_status(self, clear=True):
        """True when the flow stage's error word is raised (include/ipx.h: ipx_flow_status)."""
        r = C.c_int()
        check(lib.ipx_flow_status(self._ctx, C.byref(r), int(bool(clear))))
        return bool(r.value)

    def to_device(self, a, dtype=None):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=dtype))
        return t.to(self.device, non_blocking=False)

    def upload_frames(self, host_frames, out=None, register=True):
        """ipx_upload_frames: a C-contiguous uint8 numpy array (e.g. decoder output) -> device tensor, copied on the
        context's side stream; later calls on this engine wait for it.  ``register`` pins the array for the copy
        (ipx_host_register) unless it already is pinned; the returned handle keeps it alive and unpins it."""
        a = np.ascontiguousarray(host_frames, dtype=np.uint8)
        dev = torch.empty(a.shape, dtype=torch.uint8, device=self.device) if out is None else out
        assert dev.is_contiguous() and dev.numel() == a.size
        pinned = False
        if register:
            # (a refusal -- memory that is pinned already, or that cannot be pinned -- only means a staged copy)
            pinned = lib.ipx_host_register(C.c_void_p(a.ctypes.data), a.nbytes) == 0
        self._enter()
        check(lib.ipx_upload_frames(self._ctx, C.c_void_p(a.ctypes.data), a.nbytes, _ptr(dev)))
        return _Upload(self, dev, a, pinned)

    def _chk_frames(self, frames):
        assert frames.is_cuda and frames.device == self.device and frames.dtype == torch.uint8
        assert frames.dim() == 3 and frames.is_contiguous()
        return frames.shape

    def wpr(self, W):
        return (W + 63) // 64

    def empty_bits(self, count, H, W):
        return torch.empty((count, H, self.wpr(W)), dtype=torch.int64, device=self.device)

    def timer_begin(self):
        self._enter()
        check(lib.ipx_timer_begin(self._ctx))

    def timer_end(self):
        ms = C.c_float()
        check(lib.ipx_timer_end(self._ctx, C.byref(ms)))
        return ms.value

    # ---- built-in per-kernel profiler (HIP events on the launch stream) -----------------------------
    def profile_enable(self, on=True):
        self._enter()
        check(lib.ipx_profile_enable(self._ctx, int(bool(on))))

    def profile_reset(self):
        check(lib.ipx_profile_reset(self._ctx))

    def profile_read(self):
        """{kernel class: dict(launches, ms, bytes)} accumulated since the last reset (synchronises)."""
        out = {}
        for k in range(lib.ipx_profile_kernel_classes()):
            name, n, ms, by = C.c_char_p(), C.c_uint64(), C.c_double(), C.c_double()
            check(lib.ipx_profile_read(self._ctx, k, C.byref(name), C.byref(n), C.byref(ms), C.byref(by)))
            if n.value:
                out[name.value.decode()] = dict(launches=int(n.value), ms=ms.value, bytes=by.value)
        return out

    def profile_read_flow_levels(self):
        """{pyramid level: dict(launches, ms, bytes)} of the class "flow_iter" since the last reset (synchronises)."""
        out = {}
        for lv in range(8):
            n, ms, by = C.c_uint64(), C.c_double(), C.c_double()
            check(lib.ipx_profile_read_flow_level(self._ctx, lv, C.byref(n), C.byref(ms), C.byref(by)))
            if n.value:
                out[lv] = dict(launches=int(n.value), ms=ms.value, bytes=by.value)
        return out

    # ---- stages on device tensors ------------------------------------------------------------------
    def bgr2gray(self, bgr):
        assert bgr.dtype == torch.uint8 and bgr.is_contiguous() and bgr.shape[-1] == 3
        gray = torch.empty(bgr.shape[:-1], dtype=torch.uint8, device=self.device)
        self._enter()
        check(lib.ipx_bgr2gray(self._ctx, _ptr(bgr), gray.numel(), _ptr(gray)))
        return gray

    def unpack_dib(self, raw, rec_bytes, n, H, W, bits_per_pixel, stride, bottom_up, out=None):
        """ipx_unpack_dib: ``raw`` = device bytes of n AVI frame records (frame i at i * rec_bytes) -> (n,H,W) uint8 frames."""
        assert raw.is_cuda and raw.dtype == torch.uint8 and raw.is_contiguous()
        assert raw.numel() >= (n - 1) * rec_bytes + stride * H
        frames = torch.empty((n, H, W), dtype=torch.uint8, device=self.device) if out is None else out
        self._enter()
        check(lib.ipx_unpack_dib(self._ctx, _ptr(raw), rec_bytes, n, H, W, bits_per_pixel, stride, int(bool(bottom_up)), _ptr(frames)))
        return frames

    def flow_metric(self, frames, prev_frame=None, want_keep=True):
        n, H, W = self._chk_frames(frames)
        flows = torch.empty(n, dtype=torch.float32, device=self.device)
        keep = torch.empty(n, dtype=torch.uint8, device=self.device) if want_keep else None
        self._enter()
        check(lib.ipx_flow_metric(self._ctx, _ptr(frames), n, H, W, _ptr(prev_frame), _ptr(flows), _ptr(keep)))
        return flows, keep

    def reserve_flow(self, n, H, W):
        """Make the flow stage's plan and large scratch arenas for chunks of up to n frames now (ipx_reserve_flow)."""
        self._enter()
        check(lib.ipx_reserve_flow(self._ctx, int(n), int(H), int(W)))

    def farneback_pair(self, prev, nxt):
        H, W = prev.shape
        flow = torch.empty((H, W, 2), dtype=torch.float32, device=self.device)
        self._enter()
        check(lib.ipx_farneback_pair(self._ctx, _ptr(prev), _ptr(nxt), H, W, _ptr(flow)))
        return flow

    def background(self, frames, index):
        n, H, W = self._chk_frames(frames)
        bg = torch.empty((H, W), dtype=torch.float32, device=self.device)
        self._enter()
        check(lib.ipx_background(self._ctx, _ptr(frames), n, H, W, int(index), _ptr(bg)))
        return bg

    def damaged_mask(self, frame, bg):
        H, W = frame.shape
        assert frame.dtype == torch.uint8 and bg.shape == frame.shape
        f64 = bg.dtype == torch.float64
        assert f64 or bg.dtype == torch.float32
        mask = torch.empty((H, W), dtype=torch.uint8, device=self.device)
        thr = torch.empty_like(bg)
        self._enter()
        check(lib.ipx_damaged_mask(self._ctx, _ptr(frame), H, W, _ptr(bg), int(f64), _ptr(mask), _ptr(thr)))
        return mask, thr

    def detect_raw(self, frames, keep=None, first=0, count=None, want_dense=True, raw_out=None):
        n, H, W = self._chk_frames(frames)
        count = n - first if count is None else count
        raw = self.empty_bits(count, H, W) if raw_out is None else raw_out
        bright = self.empty_bits(count, H, W)
        dense = torch.empty((count, 2), dtype=torch.int64, device=self.device) if want_dense else None
        self._enter()
        check(lib.ipx_detect_raw(self._ctx, _ptr(frames), n, H, W, first, count, _ptr(keep), _ptr(raw), _ptr(bright),
                                 _ptr(dense)))
        return raw, bright, dense

    def remove_bright_regions(self, bright_bits, damaged_bits, H, W, out=None):
        count = damaged_bits.shape[0]
        out = torch.empty_like(damaged_bits) if out is None else out
        self._enter()
        check(lib.ipx_remove_bright_regions(self._ctx, _ptr(bright_bits), _ptr(damaged_bits), count, H, W, _ptr(out)))
        return out

    def remove_bright_regions_noalloc(self, bright_bits, damaged_bits, H, W, out):
        """ipx_remove_bright_regions_noalloc: the caller supplies ``out``; nothing is allocated."""
        self._enter()
        check(lib.ipx_remove_bright_regions_noalloc(self._ctx, _ptr(bright_bits), _ptr(damaged_bits),
                                                    damaged_bits.shape[0], H, W, _ptr(out)))
        return out

    def persistent_mask(self, raw_bits, keep, H, W):
        n = raw_bits.shape[0]
        out = self.empty_bits(1, H, W)[0]
        self._enter()
        check(lib.ipx_persistent_mask(self._ctx, _ptr(raw_bits), _ptr(keep), n, H, W, _ptr(out)))
        return out

    def or_planes(self, planes, H, W):
        out = self.empty_bits(1, H, W)[0]
        self._enter()
        check(lib.ipx_or_planes(self._ctx, _ptr(planes), planes.shape[0], H, W, _ptr(out)))
        return out

    def static_maps(self, frames, raw_bits, pers_bits, keep):
        n, H, W = self._chk_frames(frames)
        maps = torch.empty((2, H, W), dtype=torch.int32, device=self.device)
        self._enter()
        check(lib.ipx_static_maps(self._ctx, _ptr(frames), _ptr(raw_bits), _ptr(pers_bits), _ptr(keep), n, H, W,
                                  _ptr(maps[0]), _ptr(maps[1])))
        return maps

    def static_from_maps(self, maps):
        _, H, W = maps.shape
        out = self.empty_bits(1, H, W)[0]
        self._enter()
        check(lib.ipx_static_from_maps(self._ctx, _ptr(maps[0]), _ptr(maps[1]), H, W, _ptr(out)))
        return out

    def finalize(self, frames, raw_bits, pers_bits, static_bits, keep, want_bits=True):
        n, H, W = self._chk_frames(frames)
        fin = self.empty_bits(n, H, W) if want_bits else None
        counts = torch.empty((n, 3), dtype=torch.int64, device=self.device)
        self._enter()
        check(lib.ipx_finalize(self._ctx, _ptr(frames), _ptr(raw_bits), _ptr(pers_bits), _ptr(static_bits), _ptr(keep),
                               n, H, W, _ptr(fin), _ptr(counts)))
        return fin, counts

    def dense_counts(self, frames):
        n, H, W = self._chk_frames(frames)
        counts = torch.empty((n, 2), dtype=torch.int64, device=self.device)
        self._enter()
        check(lib.ipx_dense_counts(self._ctx, _ptr(frames), n, H, W, _ptr(counts)))
        return counts

    def cluster_stats(self, frames, mask_bits, keep=None, want_cleaned=False):
        n, H, W = self._chk_frames(frames)
        stats = torch.empty((n, 3), dtype=torch.int64, device=self.device)
        cleaned = self.empty_bits(n, H, W) if want_cleaned else None
        self._enter()
        check(lib.ipx_cluster_stats(self._ctx, _ptr(frames), _ptr(mask_bits), _ptr(keep), n, H, W, _ptr(self._pow2),
                                    self._pow2_dim, _ptr(stats), _ptr(cleaned)))
        return stats, cleaned

    def pack_mask(self, masks):
        """(count,H,W) u8/bool device tensor -> bit-planes."""
        count, H, W = masks.shape
        m8 = masks.to(torch.uint8).contiguous()
        bits = self.empty_bits(count, H, W)
        self._enter()
        check(lib.ipx_pack_mask(self._ctx, _ptr(m8), count, H, W, _ptr(bits)))
        return bits

    def unpack_mask(self, bits, H, W):
        count = bits.shape[0]
        out = torch.empty((count, H, W), dtype=torch.uint8, device=self.device)
        self._enter()
        check(lib.ipx_unpack_mask(self._ctx, _ptr(bits), count, H, W, _ptr(out)))
        return out

    # ---- the whole chunk (ip.py:526-632) -----------------------------------------------------------
    def detect_chunk_device(self, frames, want_final_bits=False):
        """Enqueue the whole path for a resident chunk; returns device tensors (no host sync)."""
        n, H, W = self._chk_frames(frames)
        flows = torch.empty(n, dtype=torch.float32, device=self.device)
        keep = torch.empty(n, dtype=torch.uint8, device=self.device)
        counts = torch.empty((n, 5), dtype=torch.int64, device=self.device)
        cluster = torch.empty((n, 3), dtype=torch.int64, device=self.device)
        fin = self.empty_bits(n, H, W) if want_final_bits else None
        self._enter()
        check(lib.ipx_detect_chunk(self._ctx, _ptr(frames), n, H, W, _ptr(self._pow2), self._pow2_dim, _ptr(flows),
                                   _ptr(keep), _ptr(counts), _ptr(cluster), _ptr(fin)))
        return dict(flows=flows, keep=keep, counts=counts, cluster=cluster, final_bits=fin)

    def detect_chunk_async(self, frames):
        """Enqueue the whole path and the copy-back of its per-frame integers (on a side stream into pinned
        host memory); returns a handle whose ``result()`` waits for THIS chunk only, so a driver can submit
        chunk k+1 before it post-processes chunk k (the GPU never idles on the host's float glue)."""
        if not torch.is_tensor(frames):
            frames = self.to_device(frames, np.uint8)
        return _PendingChunk(self, frames)

    def detect_chunk(self, frames, want_masks=False):
        """frames: (n,H,W) uint8 numpy array or device tensor -> dict with the four float64 arrays
        of ip.py:632 (``total_counts, cluster_counts, avg_sizes, avg_brightnesses``), ``flows``
        (float32 metric as float64, ip.py:872) and ``keep``; ``final_masks`` on request."""
        if not torch.is_tensor(frames):
            frames = self.to_device(frames, np.uint8)
        n, H, W = frames.shape
        d = self.detect_chunk_device(frames, want_final_bits=want_masks)
        keep = d["keep"].cpu().numpy()
        counts = d["counts"].cpu().numpy()
        cluster = d["cluster"].cpu().numpy()
        flows = _check_flows(d["flows"].cpu().numpy().astype(np.float64))
        total, cc, cs, cb = hostmath.chunk_results(keep, counts, cluster, H * W)
        out = dict(total_counts=total, cluster_counts=cc, avg_sizes=cs, avg_brightnesses=cb, flows=flows, keep=keep,
                   counts=counts, cluster=cluster)
        if want_masks:
            m = self.unpack_mask(d["final_bits"], H, W).cpu().numpy().astype(bool)
            out["final_masks"] = [m[i] if keep[i] else None for i in range(n)]
        return out


class Group:
    """ipx_group_*: one chunk sharded by frame range across several GPUs INSIDE libipx (one host thread, context and
    stream per device; exchanges over RCCL or peer copies).  The C-level counterpart of sharding.py, for callers
    that are one process: ``Group([0, 1, 2, 3]).detect_chunk(frames)`` returns what ``Engine.detect_chunk`` returns
    for the whole chunk.  ``devices`` may repeat a device with transport "peer" (tests on a one-GPU box)."""

    TRANSPORTS = {"auto": 0, "peer": 1, "rccl": 2}

    def __init__(self, devices, params=None, transport="auto"):
        if not torch.cuda.is_available():
            raise RuntimeError("image_processing_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        devs = [int(d) for d in devices]
        arr = (C.c_int * len(devs))(*devs)
        self._g = C.c_void_p()
        check(lib.ipx_group_create(arr, len(devs), self.TRANSPORTS[transport], C.byref(self._g)))
        nd, tr = C.c_int(), C.c_int()
        check(lib.ipx_group_info(self._g, C.byref(nd), C.byref(tr)))
        self.devices, self.transport = devs, {v: k for k, v in self.TRANSPORTS.items()}[tr.value]
        self.params = None
        self._pending = []                                 # chunks in flight (their host arrays stay alive here)
        self.set_params(params)

    def set_params(self, params=None):
        p = _ipx.default_params()
        d = dict(params or {})
        d.pop("number_of_plots", None)
        for k, v in d.items():
            if not hasattr(p, k):
                raise KeyError("unknown setting %r" % (k,))
            setattr(p, k, v)
        self.params = p
        check(lib.ipx_group_set_params(self._g, C.byref(p)))
        self._pow2, self._pow2_dim = hostmath.pow2_table(int(p.max_cluster_size))

    def close(self):
        if self._g:
            lib.ipx_group_destroy(self._g)                 # waits for whatever is still in flight
            self._g = C.c_void_p()
            self._pending = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, frames):
        """ipx_group_submit: start a chunk ((n,H,W) uint8 numpy array in HOST memory; pinned memory -- a torch pinned
        tensor's ``.numpy()``, or an array registered with ipx_host_register -- lets the upload run under the chunk before);
        returns a handle for ``wait``.  At most two chunks in flight, waited for in the order submitted."""
        a = np.ascontiguousarray(frames, dtype=np.uint8)
        n, H, W = a.shape
        h = dict(frames=a, n_pixels=H * W, flows=np.empty(n, dtype=np.float32), keep=np.empty(n, dtype=np.uint8),
                 counts=np.empty((n, 5), dtype=np.int64), cluster=np.empty((n, 3), dtype=np.int64),
                 tab=np.ascontiguousarray(self._pow2, dtype=np.float64))
        vp = lambda x: C.c_void_p(x.ctypes.data)
        check(lib.ipx_group_submit(self._g, vp(a), n, H, W, vp(h["tab"]), self._pow2_dim, vp(h["flows"]), vp(h["keep"]),
                                   vp(h["counts"]), vp(h["cluster"])))
        self._pending.append(h)
        return h

    def wait(self, handle=None):
        """ipx_group_wait for the oldest chunk in flight (``handle``, if given, must be that chunk's) -> the dict of
        Engine.detect_chunk (without masks)."""
        if not self._pending:
            raise RuntimeError("no chunk in flight")
        h = self._pending.pop(0)
        if handle is not None and handle is not h:
            self._pending.insert(0, h)
            raise RuntimeError("chunks are waited for in the order they were submitted")
        check(lib.ipx_group_wait(self._g))
        total, cc, cs, cb = hostmath.chunk_results(h["keep"], h["counts"], h["cluster"], h["n_pixels"])
        return dict(total_counts=total, cluster_counts=cc, avg_sizes=cs, avg_brightnesses=cb,
                    flows=_check_flows(h["flows"].astype(np.float64)), keep=h["keep"], counts=h["counts"], cluster=h["cluster"])

    def detect_chunk(self, frames):
        """frames: (n,H,W) uint8 numpy array in host memory -> the dict of Engine.detect_chunk (without masks)."""
        return self.wait(self.submit(frames))


class _Upload:
    """A pending ipx_upload_frames: ``tensor`` may be handed to any engine method right away (they wait on the
    device); ``release()`` waits for the copy and unpins the host array.  The handle keeps the host array alive;
    dropping it without ``release()`` (or leaving a ``with`` block) releases it the same way, so the array is never
    freed while it is registered or while the copy may still read it."""

    def __init__(self, eng, tensor, host, pinned):
        self.eng, self.tensor, self.host, self.pinned = eng, tensor, host, pinned

    def release(self):
        if self.host is None:
            return
        try:
            self.eng.synchronize()                       # the side-stream copy has read the array
        finally:
            if self.pinned:
                lib.ipx_host_unregister(C.c_void_p(self.host.ctypes.data))
                self.pinned = False
            self.host = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()
        return False

    def __del__(self):
        try:
            self.release()
        except Exception:                                # noqa: BLE001 - interpreter shutdown, context already gone
            pass


def _check_flows(flows):
    """libipx reports a failure inside the flow stage's kernels as NaN metrics (include/ipx.h, ipx_flow_metric);
    uint8 footage cannot produce a NaN metric otherwise."""
    if np.isnan(flows).any():
        raise RuntimeError("libipx: the flow stage failed on the GPU (a hand-off inside flow_iter_kernel timed out); "
                           "no result for this chunk")
    return flows


class _PendingChunk:
    """The chunk's kernels are enqueued by the constructor; ``result()`` waits for them and fetches the per-frame integers.
    The copy-back is enqueued only THEN, not behind the kernels at submit time: a device-to-host copy that sits in a DMA
    queue waiting for 50 ms of kernels holds up the next chunks' UPLOADS queued on the same engine behind it (measured in
    the chunk driver: one chunk in three started 17 ms late, the copy time of its frames; tools/avi_modes.py trace)."""

    def __init__(self, eng, frames):
        n, H, W = frames.shape
        self.n_pixels = H * W
        self.dev = eng.detect_chunk_device(frames)
        self.frames = frames                               # keep inputs and outputs alive until the copies land
        self.eng = eng
        if getattr(eng, "_copy_stream", None) is None:
            eng._copy_stream = torch.cuda.Stream(eng.device)
        self.done = torch.cuda.Event()
        self.done.record(torch.cuda.current_stream(eng.device))
        self.host = None

    def _fetch(self):
        self.done.synchronize()                            # (the stream goes on with the next chunk's kernels meanwhile)
        self.host = {}
        with torch.cuda.stream(self.eng._copy_stream):
            for k in ("keep", "counts", "cluster", "flows"):
                t = self.dev[k]
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                self.host[k] = h
            ready = torch.cuda.Event()
            ready.record(self.eng._copy_stream)
        ready.synchronize()

    def result(self):
        if self.host is None:
            self._fetch()
        keep = self.host["keep"].numpy()
        counts = self.host["counts"].numpy()
        cluster = self.host["cluster"].numpy()
        total, cc, cs, cb = hostmath.chunk_results(keep, counts, cluster, self.n_pixels)
        out = dict(total_counts=total, cluster_counts=cc, avg_sizes=cs, avg_brightnesses=cb,
                   flows=_check_flows(self.host["flows"].numpy().astype(np.float64)), keep=keep.copy(), counts=counts.copy(),
                   cluster=cluster.copy())
        self.dev = self.frames = self.eng = None
        return out

"""image_processing -- drop-in for Ben-Folkard/image_processing's module of the same name, with the
per-frame damaged-pixel detection path running as gfx950 HIP kernels (libipx.so, include/ipx.h).

Same module-level names, positional / keyword signatures and return conventions as the reference
(``ip.py`` = /root/reference/image_processing.py; each function cites the lines it mirrors), plus
the call forms of the newer API generation that the reference's own test file uses (SURVEY.md
section 4).  Frames are uint8 greyscale footage: the GPU kernels compute on the integer pixel
values, so float arrays are accepted only when they hold integers in [0, 255] (which is what the
reference's ``astype`` calls produce); anything else raises ``ValueError`` -- there is no CPU path.

``detect_damaged_pixels`` runs the whole chunk in one library call (``ipx_detect_chunk``) while the
stage functions below are the module's own; if a caller has replaced any of them (the reference's
tests monkeypatch stages), it runs stage by stage through the module globals exactly in the
reference's order instead.
"""
import math
import os

import numpy as np

try:                                    # decode only; optional (ip.py:18 imports it unconditionally)
    import cv2
except ImportError:                     # pragma: no cover - depends on the environment
    cv2 = None

try:
    import matplotlib.pyplot as plt
    import matplotlib.patches as mpatches
    HAS_MATPLOTLIB = True
except ImportError:                     # ip.py:26-33
    HAS_MATPLOTLIB = False
    plt = None
    mpatches = None

from image_processing_amd import avi as _avi
from image_processing_amd import hostmath as _hostmath

import threading as _threading

_TLS = _threading.local()


def _engine():
    """This thread's libipx context on the current torch device (created on first use).  libipx wants one
    context per (device, host thread); the reference calls its stages from worker threads (ip.py:342, 870)
    and the chunk driver below decodes on a second thread."""
    eng = getattr(_TLS, "engine", None)
    if eng is None:
        import torch
        from image_processing_amd.engine import Engine
        eng = _TLS.engine = Engine(torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return eng


class _Scoped:
    """Set engine parameters for one call, restore the previous ones afterwards."""

    def __init__(self, **params):
        self.params = params

    def __enter__(self):
        self.eng = _engine()
        self.saved = {k: getattr(self.eng.params, k) for k, _ in self.eng.params._fields_ if k != "reserved0"}
        merged = dict(self.saved)
        merged.update(self.params)
        self.eng.set_params(merged)
        return self.eng

    def __exit__(self, *exc):
        self.eng.set_params(self.saved)
        return False


def _as_u8(a, what="frames"):
    """uint8 view of footage given as any dtype holding integers in [0, 255]."""
    a = np.asarray(a)
    if a.dtype == np.uint8:
        return np.ascontiguousarray(a)
    if a.dtype == bool:
        return np.ascontiguousarray(a, dtype=np.uint8)
    r = a.astype(np.uint8)
    if not np.array_equal(r, a):
        raise ValueError("%s must hold integer values in [0, 255] (8-bit footage); the HIP path has no "
                         "float-image mode" % what)
    return np.ascontiguousarray(r)


def _stack_masks(masks, shape=None):
    """list of bool arrays / None -> (keep u8 (n,), stacked bool (n,H,W))."""
    if shape is None:
        shape = next(np.asarray(m).shape for m in masks if m is not None)
    keep = np.array([m is not None for m in masks], np.uint8)
    st = np.zeros((len(masks),) + tuple(shape), bool)
    for i, m in enumerate(masks):
        if m is not None:
            st[i] = np.asarray(m).astype(bool)
    return keep, st


def _bits(eng, masks_bool):
    import torch
    return eng.pack_mask(torch.from_numpy(np.ascontiguousarray(masks_bool)).to(eng.device))


def _unbits(eng, bits, H, W):
    if bits.dim() == 2:
        bits = bits[None]
    return eng.unpack_mask(bits, H, W).cpu().numpy().astype(bool)


# ---------------------------------------------------------------------------------------------
# video input (ip.py:38-85, 137-149)
# ---------------------------------------------------------------------------------------------
def download_video_from_url(url, filename):
    """ip.py:38-53."""
    import requests
    response = requests.get(url)
    if response.status_code == 200:
        with open(filename, "wb") as f:
            f.write(response.content)
        print(f"Downloaded video as {filename}")
    else:
        print("Failed to download video")
    return filename


def _open_capture(filename):
    if cv2 is not None:
        return cv2.VideoCapture(filename), cv2.CAP_PROP_POS_FRAMES
    return _avi.VideoCapture(filename), _avi.VideoCapture.CAP_PROP_POS_FRAMES


def _to_grey(frame):
    """cv2.cvtColor(frame, COLOR_BGR2GRAY) (ip.py:79) on the GPU: ipx_bgr2gray."""
    eng = _engine()
    return eng.bgr2gray(eng.to_device(np.ascontiguousarray(frame[..., :3], dtype=np.uint8))).cpu().numpy()


def load_video_frames(filename, frames_start=None, frames_end=None, grayscale=True):
    """ip.py:56-85: list of frames [frames_start, frames_end) as 2-D uint8 greyscale arrays.
    Decoding is cv2.VideoCapture when cv2 is importable, else the uncompressed-AVI reader."""
    cap, pos_prop = _open_capture(filename)
    frames = []
    if frames_start:
        cap.set(pos_prop, frames_start)
    frame_idx = frames_start or 0
    # our own reader hands 8-bit streams over as they are stored (cv2 would expand them to grey-coded BGR and
    # the luma conversion would give the same bytes back)
    direct = grayscale and getattr(cap, "is_gray", lambda: False)()
    pending = []                                        # positions in `frames` of colour frames not yet converted

    def convert_pending():
        # one upload / kernel / download for a batch of decoded colour frames instead of one per frame
        if pending:
            grey = _to_grey(np.stack([frames[k] for k in pending]))
            for j, k in enumerate(pending):
                frames[k] = grey[j]
            del pending[:]

    while cap.isOpened():
        if frames_end and frame_idx >= frames_end:
            break
        ret, frame = cap.read_gray() if direct else cap.read()
        if not ret:
            break
        frames.append(frame)
        if grayscale and len(frame.shape) == 3:
            if pending and frame.shape != frames[pending[0]].shape:
                convert_pending()
            pending.append(len(frames) - 1)
            if len(pending) == 32:
                convert_pending()
        frame_idx += 1
    convert_pending()
    cap.release()
    return frames


def get_video_frames_from_url(url, local_filename="temp_video.avi", frames_start=None, frames_end=None):
    """ip.py:137-149."""
    download_video_from_url(url, local_filename)
    return load_video_frames(local_filename, frames_start, frames_end)


# ---------------------------------------------------------------------------------------------
# settings (ip.py:154-180)
# ---------------------------------------------------------------------------------------------
def _prepare_settings(params):
    """ip.py:154-180: the nine defaults, overridden by ``params``, as attributes of a class ``S``."""
    values = dict(_hostmath.DEFAULT_SETTINGS)
    if params:
        values.update(params)
    return type("S", (), values)


def _engine_params(settings):
    keys = ("consecutive_threshold", "brightness_threshold", "flow_threshold", "static_threshold",
            "min_cluster_size", "max_cluster_size", "min_circularity", "sliding_window_radius")
    return {k: getattr(settings, k) for k in keys}


# ---------------------------------------------------------------------------------------------
# background (ip.py:183-252)
# ---------------------------------------------------------------------------------------------
def compute_background(frames, index=None, radius=None):
    """ip.py:183-200: background of frame ``index`` from the +-radius temporal window, centre excluded.
    Newer call form ``compute_background(frames, radius=r)``: the backgrounds of all frames, stacked."""
    if radius is None or index is None:                 # batch form: (frames, radius) or (frames, radius=r)
        radius = radius if radius is not None else (1 if index is None else index)
        return np.stack([compute_background(frames, i, radius) for i in range(len(frames))])
    start = max(0, index - radius)
    end = min(len(frames), index + radius + 1)
    window = _as_u8(np.asarray(frames[start:end]))
    if len(window) < 2:
        raise ValueError("the background window of a frame needs at least one neighbour")
    with _Scoped(sliding_window_radius=max(int(radius), 1)) as eng:
        return eng.background(eng.to_device(window), index - start).cpu().numpy()


def _find_background(frames, pixel_std_coeff=1.0):
    """ip.py:203-252 on a stack of neighbour frames (centre already removed): mean of the values
    <= mean + coeff*std per pixel, float32."""
    stack = _as_u8(frames)
    n = len(stack)
    if not 1 <= n <= 16:
        raise ValueError("_find_background handles 1..16 neighbour frames on the GPU (window radius <= 8)")
    k = n // 2                                          # dummy centre in the middle; window = all n others, in order
    ext = np.concatenate([stack[:k], stack[:1], stack[k:]])
    with _Scoped(sliding_window_radius=max(k, n - k), pixel_std_coeff=float(pixel_std_coeff)) as eng:
        return eng.background(eng.to_device(ext), k).cpu().numpy()


# ---------------------------------------------------------------------------------------------
# raw mask (ip.py:255-267, 635-648)
# ---------------------------------------------------------------------------------------------
def get_damaged_pixel_mask(frame, *args):
    """ip.py:635-648 -> (uint8 mask, threshold array).  Accepts ``(frame, height, width, background)``
    (this reference) and ``(frame, background)`` (newer form)."""
    background = args[-1]
    eng = _engine()
    f8 = _as_u8(frame, "frame")
    bg = np.asarray(background)
    if bg.dtype != np.float32:
        bg = bg.astype(np.float64)                      # ip.py:638 follows the background's dtype
    m, thr = eng.damaged_mask(eng.to_device(f8), eng.to_device(bg))
    return m.cpu().numpy(), thr.cpu().numpy()


def _raw_damaged_mask(frame, background):
    """ip.py:255-267."""
    mask, _ = get_damaged_pixel_mask(frame, background)
    return np.asarray(mask).astype(bool)


# ---------------------------------------------------------------------------------------------
# temporal maps (ip.py:270-310, 438-481)
# ---------------------------------------------------------------------------------------------
def compute_persistent_mask(masks, consecutive_threshold):
    """ip.py:270-289: pixels flagged in >= ``consecutive_threshold`` consecutive frames; ``None`` resets."""
    H, W = np.asarray(masks[0]).shape                   # ip.py:275 (frame 0 is always kept)
    keep, st = _stack_masks(masks, (H, W))
    with _Scoped(consecutive_threshold=int(consecutive_threshold)) as eng:
        p = eng.persistent_mask(_bits(eng, st), eng.to_device(keep), H, W)
        return _unbits(eng, p, H, W)[0]


def _suppress(masks, bad_planes):
    """m & ~(OR of bad_planes) for every non-None mask (ipx_finalize) -> (list, counts list)."""
    shape = next((np.asarray(m).shape for m in masks if m is not None), None)
    if shape is None:
        return [None] * len(masks), [np.nan] * len(masks)
    H, W = shape
    keep, st = _stack_masks(masks, shape)
    eng = _engine()
    planes = [np.asarray(b).astype(bool) for b in bad_planes]
    pers = _bits(eng, planes[0][None])[0]
    stat = _bits(eng, planes[1][None])[0] if len(planes) > 1 else None
    import torch
    dummy = torch.zeros((len(masks), H, W), dtype=torch.uint8, device=eng.device)
    fin, counts = eng.finalize(dummy, _bits(eng, st), pers, stat, eng.to_device(keep))
    fin = _unbits(eng, fin, H, W)
    counts = counts.cpu().numpy()
    out = [fin[i] if keep[i] else None for i in range(len(masks))]
    cnt = [int(counts[i, 0]) if keep[i] else np.nan for i in range(len(masks))]
    return out, cnt


def filter_consecutive_pixels(masks, persistent, return_counts=True):
    """ip.py:292-310."""
    out, cnt = _suppress(masks, [persistent])
    return (out, cnt) if return_counts else out


def compute_static_mask(frames, masks, brightness_threshold, static_threshold, min_valid_frames):
    """ip.py:438-465."""
    fr = _as_u8(np.stack([np.asarray(f) for f in frames]))
    H, W = fr.shape[1:]
    keep, st = _stack_masks(masks, (H, W))
    with _Scoped(brightness_threshold=float(brightness_threshold), static_threshold=float(static_threshold),
                 min_valid_frames=int(min_valid_frames)) as eng:
        zero = eng.empty_bits(1, H, W)[0].zero_()
        maps = eng.static_maps(eng.to_device(fr), _bits(eng, st), zero, eng.to_device(keep))
        return _unbits(eng, eng.static_from_maps(maps), H, W)[0]


def apply_static_suppression(masks, persistent, static_mask, calc_counts=True):
    """ip.py:468-481."""
    out, cnt = _suppress(masks, [persistent, static_mask])
    return (out, np.array(cnt, dtype=float)) if calc_counts else out


# ---------------------------------------------------------------------------------------------
# bright-area estimates and final counts (ip.py:351-365, 770-847)
# ---------------------------------------------------------------------------------------------
def _frame_counters(frames, masks, brightness_threshold, estimate_threshold):
    """Per frame (keep, [sum m, #(m & f<est), #(m & f>bt), #(f<est), #(f>bt)])."""
    fr = _as_u8(np.stack([np.asarray(f) for f in frames]))
    n, H, W = fr.shape
    keep, st = _stack_masks(masks, (H, W))
    with _Scoped(brightness_threshold=float(brightness_threshold),
                 estimate_threshold=int(math.ceil(estimate_threshold))) as eng:
        fd = eng.to_device(fr)
        zero = eng.empty_bits(1, H, W)[0].zero_()
        _, c3 = eng.finalize(fd, _bits(eng, st), zero, None, None, want_bits=False)
        d2 = eng.dense_counts(fd)
        return keep, np.concatenate([c3.cpu().numpy(), d2.cpu().numpy()], axis=1), H * W


def estimate_damaged_pixels_in_bright_areas(frames, damaged_pixel_masks, brightness_threshold=170):
    """ip.py:770-821: damaged-pixel density of the dark area times the bright area, per frame."""
    _, c, P = _frame_counters(frames, damaged_pixel_masks, brightness_threshold, brightness_threshold)
    out = np.full(len(c), np.nan, np.float64)
    for i, (m, m_low, _, n_low, _) in enumerate(c.tolist()):
        low = n_low - m_low
        high = (P - n_low) - (m - m_low)
        if low > 0:
            out[i] = round((m / low) * high)
    return out


def find_bright_area_estimates(frames, damaged_pixel_masks, brightness_threshold):
    """ip.py:824-847 (the inner estimate uses the default 170, ip.py:836)."""
    keep, c, P = _frame_counters(frames, damaged_pixel_masks, brightness_threshold, 170)
    out = np.full(len(c), np.nan, np.float64)
    for i, row in enumerate(c.tolist()):
        if keep[i]:
            out[i] = _hostmath.bright_area_estimate(*row, P)
    return out


def _get_final_count(masks, bright_estimates):
    """ip.py:351-365."""
    shape = next((np.asarray(m).shape for m in masks if m is not None), None)
    if shape is None:
        return np.array([np.nan + (e if not np.isnan(e) else 0) for e in bright_estimates], dtype=float)
    keep, st = _stack_masks(masks, shape)
    eng = _engine()
    import torch
    H, W = shape
    dummy = torch.zeros((len(masks), H, W), dtype=torch.uint8, device=eng.device)
    zero = eng.empty_bits(1, H, W)[0].zero_()
    _, c3 = eng.finalize(dummy, _bits(eng, st), zero, None, None, want_bits=False)
    sums = c3.cpu().numpy()[:, 0]
    return np.array([(sums[i] if keep[i] else np.nan) + (e if not np.isnan(e) else 0)
                     for i, e in enumerate(bright_estimates)], dtype=float)


# ---------------------------------------------------------------------------------------------
# cluster statistics (ip.py:332-348, 651-728)
# ---------------------------------------------------------------------------------------------
def filter_damaged_pixel_clusters(frame, damaged_pixel_mask, min_cluster_size, max_cluster_size, min_circularity,
                                  circularity_size_threshold=10):
    """ip.py:651-728 -> (cleaned bool mask, cluster count, mean cluster size, mean cluster brightness).
    A structuring element passed in the 6th position (newer API) is accepted and must be the 3x3 cross."""
    if isinstance(circularity_size_threshold, np.ndarray):
        if not np.array_equal(circularity_size_threshold != 0, [[0, 1, 0], [1, 1, 1], [0, 1, 0]]):
            raise ValueError("only the 3x3 cross (cv2 MORPH_ELLIPSE 3x3) is supported")
        circularity_size_threshold = 10
    f8 = _as_u8(frame, "frame")
    H, W = f8.shape
    m = np.asarray(damaged_pixel_mask).astype(bool)
    with _Scoped(min_cluster_size=int(min_cluster_size), max_cluster_size=int(max_cluster_size),
                 min_circularity=float(min_circularity),
                 circularity_size_threshold=int(circularity_size_threshold)) as eng:
        stats, cleaned = eng.cluster_stats(eng.to_device(f8[None]), _bits(eng, m[None]), want_cleaned=True)
        c, sa, sb = stats.cpu().numpy()[0].tolist()
        cleaned = _unbits(eng, cleaned, H, W)[0]
    if c == 0:
        return np.zeros((H, W), bool), 0, 0.0, float("nan")
    return cleaned, int(c), float(sa / c), float(sb / sa)


_ORIG = {}


def _pristine(*names):
    g = globals()
    return all(g[n] is _ORIG[n] for n in names)


def compute_cluster_stats(frames, masks, flows, settings, n_jobs=-1):
    """ip.py:332-348 -> array of three float64 rows (cluster_count, cluster_size, cluster_brightness)."""
    n = len(masks)
    if not _pristine("filter_damaged_pixel_clusters"):                       # a caller replaced the stage
        # Newer API generation (t.py:388-419): the settings carry the structuring element, which is the stage's
        # 6th positional argument, and a discarded frame is an all-False mask instead of None.
        extra = (settings.kernel,) if hasattr(settings, "kernel") else ()
        res = []
        for frame, mask, flow in zip(frames, masks, flows):
            if mask is None or flow > settings.flow_threshold or (extra and not np.asarray(mask).any()):
                res.append((0.0, 0.0, 0.0))
            else:
                _, c, s, b = filter_damaged_pixel_clusters(frame, mask, settings.min_cluster_size,
                                                           settings.max_cluster_size, settings.min_circularity, *extra)
                res.append((c, s, b))
        return np.array(res).T
    res = np.zeros((n, 3), np.float64)
    live = [i for i in range(n) if not (masks[i] is None or flows[i] > settings.flow_threshold)]
    if live:
        fr = _as_u8(np.stack([np.asarray(frames[i]) for i in live]))
        ms = np.stack([np.asarray(masks[i]).astype(bool) for i in live])
        with _Scoped(min_cluster_size=int(settings.min_cluster_size), max_cluster_size=int(settings.max_cluster_size),
                     min_circularity=float(settings.min_circularity)) as eng:
            stats, _ = eng.cluster_stats(eng.to_device(fr), _bits(eng, ms))
            stats = stats.cpu().numpy().tolist()
        for i, (c, sa, sb) in zip(live, stats):
            res[i] = (c, sa / c, sb / sa) if c else (0, 0.0, np.nan)
    return res.T


# ---------------------------------------------------------------------------------------------
# bright-region rejection (ip.py:731-767)
# ---------------------------------------------------------------------------------------------
def remove_bright_regions(background, brightness_threshold, filtered_damaged_pixels, max_cluster_size):
    """ip.py:731-767: drop damaged components that touch a bright-background blob of >= max_cluster_size px."""
    bright = np.asarray(background) > brightness_threshold
    dmg = np.asarray(filtered_damaged_pixels).astype(bool)
    H, W = dmg.shape
    with _Scoped(max_cluster_size=int(max_cluster_size)) as eng:
        out = eng.remove_bright_regions(_bits(eng, bright[None]), _bits(eng, dmg[None]), H, W)
        return _unbits(eng, out, H, W)[0]


# ---------------------------------------------------------------------------------------------
# motion test (ip.py:852-904)
# ---------------------------------------------------------------------------------------------
def compute_optical_flow_metric(frames):
    """ip.py:852-873: flows[0] = 0, flows[i] = mean Farneback flow magnitude between frames i-1 and i."""
    if len(frames) == 0:
        return np.zeros(0)
    fr = _as_u8(np.stack([np.asarray(f) for f in frames]))
    eng = _engine()
    flows, _ = eng.flow_metric(eng.to_device(fr), want_keep=False)
    return flows.cpu().numpy().astype(np.float64)


def filter_frames_by_optical_flow(frames, pixel_counts, optical_flows, damaged_pixel_masks, threshold,
                                  removed_displayed=False):
    """ip.py:876-904 (not on the detection path; plain indexing)."""
    frames = np.asarray(frames)
    pixel_counts = np.asarray(pixel_counts)
    optical_flows = np.asarray(optical_flows)
    damaged_pixel_masks = np.asarray(damaged_pixel_masks)
    valid = optical_flows <= threshold
    if removed_displayed:
        print(f"Removed {len(frames) - int(valid.sum())} frames due to high optical flow")
    return frames[valid], pixel_counts[valid], damaged_pixel_masks[valid], optical_flows[valid]


def find_damaged_pixel_heatmap(frames, damaged_pixel_masks, brightness_threshold):
    """ip.py:907-933: heat / valid * 100 where more than 10 frames are valid (same sums as the static map)."""
    fr = _as_u8(np.asarray(frames))
    H, W = fr.shape[1:]
    ms = np.asarray(damaged_pixel_masks).astype(bool)
    with _Scoped(brightness_threshold=float(brightness_threshold)) as eng:
        zero = eng.empty_bits(1, H, W)[0].zero_()
        maps = eng.static_maps(eng.to_device(fr), _bits(eng, ms), zero, None).cpu().numpy()
    heat, valid = maps[0].astype(np.int64), maps[1].astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(valid > 10, heat / valid * 100.0, 0.0)


# ---------------------------------------------------------------------------------------------
# orchestration (ip.py:526-632)
# ---------------------------------------------------------------------------------------------
_STAGES = ("compute_optical_flow_metric", "compute_background", "_raw_damaged_mask", "get_damaged_pixel_mask",
           "remove_bright_regions", "compute_persistent_mask", "filter_consecutive_pixels", "compute_static_mask",
           "apply_static_suppression", "find_bright_area_estimates", "estimate_damaged_pixels_in_bright_areas",
           "_get_final_count", "compute_cluster_stats", "filter_damaged_pixel_clusters", "_prepare_settings")


def detect_damaged_pixels(frames, chunk_index=0, plot=False, show_plots=False, save_plots=False,
                          output_folder="results", visualise_damaged_pixels_plot=True, plot_damaged_pixels_plot=True,
                          plot_heatmap_plot=True, params=None):
    """ip.py:526-632: one chunk of greyscale frames -> (total_counts, cluster_counts, avg_sizes,
    avg_brightnesses), four float64 arrays; NaN count = frame discarded by the motion test."""
    if not _pristine(*_STAGES):
        return _detect_stagewise(frames, chunk_index, plot, show_plots, save_plots, output_folder,
                                 visualise_damaged_pixels_plot, plot_damaged_pixels_plot, plot_heatmap_plot, params)
    settings = _prepare_settings(params)
    fr = _as_u8(np.asarray(frames))
    with _Scoped(**_engine_params(settings)) as eng:
        r = eng.detect_chunk(fr, want_masks=bool(plot))
    if plot:
        _generate_plots(fr, chunk_index, r["final_masks"], r["total_counts"], r["flows"], settings,
                        show_plots=show_plots, save_plots=save_plots, output_folder=output_folder,
                        visualise_damaged_pixels_plot=visualise_damaged_pixels_plot,
                        plot_damaged_pixels_plot=plot_damaged_pixels_plot, plot_heatmap_plot=plot_heatmap_plot)
    return r["total_counts"], r["cluster_counts"], r["avg_sizes"], r["avg_brightnesses"]


def _positional_arity(fn):
    """Number of positional parameters of a replaced stage (the two API generations differ in it)."""
    import inspect
    try:
        ps = inspect.signature(fn).parameters.values()
    except (TypeError, ValueError):
        return None
    if any(q.kind == q.VAR_POSITIONAL for q in ps):
        return None
    return sum(q.kind in (q.POSITIONAL_ONLY, q.POSITIONAL_OR_KEYWORD) for q in ps)


def _detect_stagewise(frames, chunk_index, plot, show_plots, save_plots, output_folder,
                      visualise_damaged_pixels_plot, plot_damaged_pixels_plot, plot_heatmap_plot, params):
    """The same sequence, one module-global stage call at a time (used when a stage was replaced)."""
    settings = _prepare_settings(params)
    frames = np.asarray(frames)
    optical_flows = compute_optical_flow_metric(frames)
    raw_masks = np.full(len(frames), None, dtype=object)
    batch_backgrounds = None
    if not _pristine("compute_background") and _positional_arity(compute_background) == 2:
        batch_backgrounds = compute_background(frames, settings.sliding_window_radius)   # newer form, t.py:513-514
    for i, frame in enumerate(frames):
        if optical_flows[i] <= settings.flow_threshold:
            background = batch_backgrounds[i] if batch_backgrounds is not None else \
                compute_background(frames, i, settings.sliding_window_radius)
            raw_mask = _raw_damaged_mask(frame, background)
            raw_masks[i] = remove_bright_regions(background, settings.brightness_threshold, raw_mask,
                                                 settings.max_cluster_size)
    persistent_pixels = compute_persistent_mask(raw_masks, settings.consecutive_threshold)
    clean_masks, _ = filter_consecutive_pixels(raw_masks, persistent_pixels)
    static_mask = compute_static_mask(frames, clean_masks, settings.brightness_threshold, settings.static_threshold, 10)
    final_masks, total_counts = apply_static_suppression(clean_masks, persistent_pixels, static_mask)
    bright_area_estimates = find_bright_area_estimates(frames, final_masks, settings.brightness_threshold)
    total_counts = _get_final_count(final_masks, bright_area_estimates)
    cluster_counts, avg_sizes, avg_brightnesses = compute_cluster_stats(frames, final_masks, optical_flows, settings)
    if plot:
        _generate_plots(frames, chunk_index, final_masks, total_counts, optical_flows, settings,
                        show_plots=show_plots, save_plots=save_plots, output_folder=output_folder,
                        visualise_damaged_pixels_plot=visualise_damaged_pixels_plot,
                        plot_damaged_pixels_plot=plot_damaged_pixels_plot, plot_heatmap_plot=plot_heatmap_plot)
    return total_counts, cluster_counts, avg_sizes, avg_brightnesses


# ---------------------------------------------------------------------------------------------
# plots (ip.py:381-435, 936-1065, 1096-1133) -- presentation only, matplotlib optional
# ---------------------------------------------------------------------------------------------
def _is_index(v):
    return isinstance(v, (int, np.integer)) and not isinstance(v, (bool, np.bool_))


def _generate_plots(frames, *args, **kwargs):
    """ip.py:381-435: ``(frames, chunk_index, masks, counts, flows, settings, show_plots=True, ...)``.
    Newer form (t.py:433-458): ``(frames, masks, counts, flows, settings)``, and the plot functions are then
    called without a chunk index."""
    names = ("show_plots", "save_plots", "output_folder", "visualise_damaged_pixels_plot",
             "plot_damaged_pixels_plot", "plot_heatmap_plot")
    legacy = "chunk_index" in kwargs or (len(args) >= 1 and (_is_index(args[0]) or args[0] is None) and len(args) >= 5)
    if legacy:
        lead = ["chunk_index", "masks", "counts", "flows", "settings"]
    else:
        lead = ["masks", "counts", "flows", "settings"]
    if len(args) > len(lead) + len(names):
        raise TypeError("_generate_plots: too many positional arguments")
    opts = dict(show_plots=True, save_plots=False, output_folder="results", visualise_damaged_pixels_plot=True,
                plot_damaged_pixels_plot=True, plot_heatmap_plot=True)
    bound = dict(zip(lead + list(names), args))
    for k, v in kwargs.items():
        if k in bound:
            raise TypeError("_generate_plots: %s given twice" % k)
        bound[k] = v
    missing = [k for k in lead if k not in bound]
    if missing:
        raise TypeError("_generate_plots: missing %s" % ", ".join(missing))
    opts.update({k: bound[k] for k in names if k in bound})
    masks, counts, flows, settings = (bound[k] for k in ("masks", "counts", "flows", "settings"))
    chunk = (bound["chunk_index"],) if legacy else ()
    out = dict(show_plots=opts["show_plots"], save_plots=opts["save_plots"], output_folder=opts["output_folder"])
    if opts["visualise_damaged_pixels_plot"]:
        survivors = [i for i, flow in enumerate(flows) if flow <= settings.flow_threshold]
        for i in survivors[:settings.number_of_plots]:
            visualise_damaged_pixels(frames[i], i, *chunk, masks[i], counts[i], **out)
    if opts["plot_damaged_pixels_plot"]:
        plot_damaged_pixels(counts, *chunk, **out)
    if opts["plot_heatmap_plot"]:
        shape = np.asarray(frames[0]).shape
        heatmap = find_damaged_pixel_heatmap(
            frames, [np.asarray(m).astype(bool) if m is not None else np.zeros(shape, bool) for m in masks],
            settings.brightness_threshold)
        plot_heatmap(heatmap, *chunk, **out)


def _finish_figure(show_plots, save_plots, output_folder, file_name, tight_first=False):
    """Common tail of the plot functions: show and / or save the current figure as ``file_name`` (300 dpi), then close it.
    The reference lays the figure out tightly before saving; its combined overview does so before showing as well."""
    if tight_first:
        plt.tight_layout()
    if show_plots and not tight_first:
        plt.show()
    if save_plots:
        if not tight_first:
            plt.tight_layout()
        os.makedirs(output_folder, exist_ok=True)
        plt.savefig(os.path.join(output_folder, file_name), dpi=300)
    if show_plots and tight_first:
        plt.show()
    plt.close()


def _blend(base, overlay):
    """cv2.addWeighted(base, 0.8, overlay, 1.0, 0) for uint8 images."""
    return np.clip(np.rint(base.astype(np.float64) * 0.8 + overlay.astype(np.float64)), 0, 255).astype(np.uint8)


def visualise_damaged_pixels(frame, frame_index, *args, **kwargs):
    """ip.py:936-1002: the frame next to a copy with damaged pixels and bright areas overlaid.
    ``(frame, frame_index, chunk_index, cluster_mask, cluster_count, bright_threshold=170, show_plots=True,
    save_plots=False, output_folder="results")``; the newer generation omits ``chunk_index`` (t.py:443-444)."""
    legacy = "chunk_index" in kwargs or (len(args) >= 1 and _is_index(args[0]))
    lead = (["chunk_index"] if legacy else []) + ["cluster_mask", "cluster_count", "bright_threshold", "show_plots",
                                                   "save_plots", "output_folder"]
    if len(args) > len(lead):
        raise TypeError("visualise_damaged_pixels: too many positional arguments")
    b = dict(chunk_index=None, bright_threshold=170, show_plots=True, save_plots=False, output_folder="results")
    b.update(zip(lead, args))
    b.update(kwargs)
    chunk_index, cluster_mask, cluster_count = b["chunk_index"], b["cluster_mask"], b["cluster_count"]
    bright_threshold, show_plots, save_plots, output_folder = (b["bright_threshold"], b["show_plots"], b["save_plots"],
                                                               b["output_folder"])
    if not HAS_MATPLOTLIB:
        print("matplotlib not available - skipping damaged pixel visualisation")
        return
    frame = np.asarray(frame)
    vis = np.repeat(frame[:, :, None], 3, axis=2)                       # grey -> BGR
    over = np.zeros_like(vis)
    if cluster_mask is not None:
        over[np.asarray(cluster_mask).astype(bool)] = (0, 165, 255)
    vis = _blend(vis, over)
    over = np.zeros_like(vis)
    over[frame > bright_threshold] = (255, 0, 0)
    vis = _blend(vis, over)
    plt.figure(figsize=(14, 6))
    plt.subplot(1, 2, 1)
    plt.imshow(frame, cmap="gray", vmin=0, vmax=255)
    plt.title(f"original chunk {chunk_index}, frame {frame_index}")
    plt.axis("off")
    plt.subplot(1, 2, 2)
    plt.imshow(vis[:, :, ::-1])                                         # BGR -> RGB
    plt.title(f"clusters: {cluster_count}")
    plt.axis("off")
    plt.legend(handles=[mpatches.Patch(color="orange", label="Damaged Pixels"),
                        mpatches.Patch(color="royalblue", label="Bright Background Areas")],
               loc="upper left", fontsize="small", frameon=True)
    _finish_figure(show_plots, save_plots, output_folder, f"visualise_damaged_pixels_frame_{chunk_index=}_{frame_index=}.png")


def plot_heatmap(heatmap, chunk_index=None, title="Damaged Pixel Heatmap", show_plots=True, save_plots=False,
                 output_folder="results", figsize=(15, 10)):
    """ip.py:1005-1032 (``chunk_index`` is optional: the newer generation calls it without, t.py:450-451)."""
    if not HAS_MATPLOTLIB:
        print("matplotlib not available - skipping heatmap plot")
        return
    plt.figure(figsize=figsize)
    plt.imshow(heatmap, cmap="viridis", interpolation="nearest")
    plt.colorbar(label="Percentage of frames (%)")
    plt.title(f"{title} ({chunk_index = })")
    _finish_figure(show_plots, save_plots, output_folder, f"{title.lower().replace(' ', '_')}_{chunk_index=}.png")


def plot_damaged_pixels(damaged_pixel_counts, chunk_index=None, show_plots=True, save_plots=False,
                        output_folder="results"):
    """ip.py:1035-1065 (``chunk_index`` is optional: the newer generation calls it without, t.py:445-447)."""
    if not HAS_MATPLOTLIB:
        print("matplotlib not available - skipping damaged pixel output graph")
        return
    plt.figure(figsize=(10, 5))
    plt.plot(damaged_pixel_counts, label="Damaged Pixels Count", color="blue")
    plt.xlabel("Frame Number")
    plt.ylabel("Number of Damaged Pixels")
    plt.title(f"Damaged Pixels Detected Over Time ({chunk_index = })")
    plt.legend()
    _finish_figure(show_plots, save_plots, output_folder, f"plot_damged_pixels_{chunk_index=}.png")


def chunked_nanmean(array, step):
    """ip.py:1068-1085: nan-means over windows of ``step`` samples, the last window may be partial."""
    if len(array) == 0:
        return [np.nan]
    array = np.asarray(array)
    n_full = (len(array) // step) * step
    means = np.nanmean(array[:n_full].reshape(-1, step), axis=1)
    remainder = len(array) % step
    if remainder > 0:
        means = np.append(means, np.nanmean(array[-remainder:]))
    return means.tolist()


def plot_all_chunks(all_counts, all_clusters, all_sizes, all_brightness, times, show_plots=True, save_plots=False,
                    output_folder="results"):
    """ip.py:1096-1133."""
    if not HAS_MATPLOTLIB:
        print("matplotlib not available - skipping combined chunk plot")
        return
    fig, axes = plt.subplots(4, 1, figsize=(12, 10), sharex=True)
    for ax, series, colour, label in zip(axes, (all_counts, all_clusters, all_sizes, all_brightness),
                                         ("blue", "orange", "green", "red"),
                                         ("Damaged Pixel Count", "Cluster Count", "Average Cluster Size",
                                          "Average Brightness")):
        ax.plot(series, color=colour, label=label)
        ax.legend(loc="upper right")
        ax.grid(True, linestyle="--", alpha=0.5)
    axes[0].set_title("Combined Chunk Results Over Frames")
    axes[-1].set_xlabel("Frame Index" if times is None else "Time (s)")
    if times is not None:
        for ax in axes:
            ax.set_xlim([0, len(times)])
    _finish_figure(show_plots, save_plots, output_folder, "combined_chunks_overview.png", tight_first=True)


# ---------------------------------------------------------------------------------------------
# chunk driver and CLI (ip.py:1136-1350)
# ---------------------------------------------------------------------------------------------
def _video_info(video_filename):
    cap, _ = _open_capture(video_filename)
    if cv2 is not None:
        n, fps = int(cap.get(cv2.CAP_PROP_FRAME_COUNT)), cap.get(cv2.CAP_PROP_FPS)
    else:
        n, fps = int(cap.get(_avi.VideoCapture.CAP_PROP_FRAME_COUNT)), cap.get(_avi.VideoCapture.CAP_PROP_FPS)
    cap.release()
    return n, fps


_DECODE_THREADS = 8


class _RawChunk:
    """A chunk still in the layout the AVI file stores it in: the bytes of frames [start, end) -- DIBs, bottom-up rows
    padded to 4 bytes, 8 bytes of chunk header between consecutive frames -- read with a few large positional reads into
    a pinned staging buffer.  The row flip, the padding and (24-bit streams) the luma conversion are left to the GPU
    (ipx_unpack_dib), so the decode thread moves every byte exactly once, at memcpy speed."""

    def __init__(self, slot, nbytes, n, rec, height, width, bpp, stride, bottom_up, staged=None, uploaded=None):
        self.slot, self.nbytes, self.n, self.rec = slot, nbytes, n, rec
        self.height, self.width, self.bpp, self.stride, self.bottom_up = height, width, bpp, stride, bottom_up
        self.staged, self.uploaded = staged, uploaded     # the same bytes in HBM and the event of their arrival

    def __len__(self):
        return self.n

    def frames(self):
        """The frames as load_video_frames returns them (only a stage-by-stage run needs them on the host)."""
        raw = self.slot["host"].numpy()[:self.nbytes]
        out = []
        for i in range(self.n):
            a = raw[i * self.rec:i * self.rec + self.stride * self.height].reshape(self.height, self.stride)
            a = a[::-1] if self.bottom_up else a
            if self.bpp == 8:
                out.append(np.ascontiguousarray(a[:, :self.width]))
            else:
                out.append(_to_grey(np.ascontiguousarray(a[:, :3 * self.width].reshape(self.height, self.width, 3))))
        return out


class _StagingRing:
    """Three pinned staging buffers of one chunk driver: one being read into by its decode thread, one uploading, one
    spare -- so the decode of chunk k+2 can start while chunk k+1 uploads.  Buffers are pinned once and reused for every
    later chunk (pinning 2 GB of fresh memory per chunk was two thirds of a first pass's time); a driver returns its
    ring to the pool when it is done, and the next video's driver takes it over."""
    SLOTS = 3

    def __init__(self):
        self.slots, self.next = [], 0

    def take(self, nbytes):
        """The next buffer, at least ``nbytes`` long and no longer read by an upload (waits for that upload if need be)."""
        import torch
        if len(self.slots) < self.SLOTS:
            self.slots.append(dict(host=None, busy=None))
            slot = self.slots[-1]
        else:
            slot = self.slots[self.next % self.SLOTS]
        self.next += 1
        if slot["busy"] is not None:
            slot["busy"].synchronize()
            slot["busy"] = None
        if slot["host"] is None or slot["host"].numel() < nbytes:
            slot["host"] = None                           # (released before the larger one is pinned)
            slot["host"] = torch.empty(int(nbytes + (nbytes >> 4)), dtype=torch.uint8, pin_memory=True)
        return slot


_RING_POOL = []
_RING_LOCK = _threading.Lock()
#: set to a list to have the chunk drivers append (chunk index, seconds when taken up, seconds when delivered) to it
#: (time.perf_counter); bench.py reads the steady-state rate of the input side from it
_TIMELINE = None


def _ring_acquire():
    with _RING_LOCK:
        return _RING_POOL.pop() if _RING_POOL else _StagingRing()


def _ring_release(ring):
    for slot in ring.slots:                               # nothing may still be reading the buffers when the next driver writes them
        if slot["busy"] is not None:
            slot["busy"].synchronize()
            slot["busy"] = None
    with _RING_LOCK:
        if len(_RING_POOL) < 16:
            _RING_POOL.append(ring)


_UPLOAD_PIECE = 64 << 20
_UPLOAD_IN_DECODER = os.environ.get("IPX_UPLOAD_IN_DECODER", "1") != "0"   # (A/B switch: tools/avi_modes.py)

# ---- the next video's first chunk (not in the reference) ----------------------------------------------------------
# A video's first chunk is the one decode + upload that nothing overlaps (~25 ms per GB).  A driver that processes
# several videos one after the other (image_processing_amd/multi.py, bench.py --videos) names the next one with
# queue_next_video(); the chunk pipeline of the CURRENT video then starts decoding it as soon as its own decode thread
# has nothing left to read, and main(next video) finds the chunk waiting.
_NEXT_LOCK = _threading.Lock()
_NEXT_VIDEO = None             # (filename, step_size) announced and not started yet
_PREFETCHED = {}               # real path -> dict(span, device, ring, future)
_PREFETCH_POOL = None
_PREFETCH_STATS = dict(started=0, taken=0)


def queue_next_video(video_filename, step_size=1000):
    """Announce the video that main() will be called on after the one it is called on next (same thread, same device,
    this ``step_size``), so that its first chunk is decoded and uploaded under the last chunk of the current one.
    Purely an optimisation: a video that was announced and never processed, or processed with other arguments, costs
    one wasted decode.  ``None`` withdraws the announcement."""
    global _NEXT_VIDEO
    with _NEXT_LOCK:
        _NEXT_VIDEO = (str(video_filename), int(step_size)) if video_filename is not None else None


def _prefetch_start(device):
    """Called by a chunk pipeline whose decode thread has read its last chunk: start on the announced video."""
    global _NEXT_VIDEO, _PREFETCH_POOL
    with _NEXT_LOCK:
        nxt, _NEXT_VIDEO = _NEXT_VIDEO, None
        if nxt is None:
            return
        if _PREFETCH_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            _PREFETCH_POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ipx-prefetch")
    video, step_size = nxt
    try:
        num_frames, _ = _video_info(video)
        span = (0, int(min(step_size, num_frames)))
        if span[1] < 2:
            return
        ring = _ring_acquire()
        try:
            fut = _PREFETCH_POOL.submit(_load_chunk, video, span[0], span[1], ring, device)
        except BaseException:
            _ring_release(ring)
            raise
        _PREFETCH_STATS["started"] += 1
        with _NEXT_LOCK:
            old = _PREFETCHED.pop(os.path.realpath(video), None)
            _PREFETCHED[os.path.realpath(video)] = dict(span=span, device=device, ring=ring, future=fut)
        if old is not None:
            _prefetch_drop(old)
    except Exception:                                   # noqa: BLE001 - an optimisation only: main() reads the file itself
        pass


def _prefetch_drop(entry):
    try:
        entry["future"].result()
    except Exception:                                   # noqa: BLE001
        pass
    _ring_release(entry["ring"])


def _prefetch_take(video_filename, span, device):
    """(ring, future) of the chunk ``span`` of this video if it was prefetched for ``device``, else None; a prefetch of
    the same video that does not fit is waited for and dropped."""
    with _NEXT_LOCK:
        entry = _PREFETCHED.pop(os.path.realpath(str(video_filename)), None)
    if entry is None:
        return None
    if entry["span"] != tuple(span) or entry["device"] != device or not _pristine("load_video_frames"):
        _prefetch_drop(entry)
        return None
    _PREFETCH_STATS["taken"] += 1
    return entry["ring"], entry["future"]


def _load_chunk_raw_avi(video_filename, start, end, ring, device=None):
    """The decode thread's short cut for uncompressed AVI files (what the benchmarks and the camera pipeline store) read
    by our own reader, with load_video_frames not replaced by the caller: the chunk's bytes go from the page cache into
    pinned memory as they are stored, in 64 MB pieces read by a few threads, and every piece is sent on to HBM (this
    thread's copy stream on ``device``) as soon as it and the pieces before it are there -- the upload runs under the
    reads instead of after them, which is what a video's FIRST chunk, that nothing else overlaps, is waiting for.
    Returns None when any of that does not hold (cv2 importable, compressed or palettised-colour stream, frames not
    evenly spaced in the file)."""
    if ring is None or cv2 is not None or not _pristine("load_video_frames"):
        return None
    try:
        import torch
        if not torch.cuda.is_available():
            return None
        reader = _avi.AviReader(video_filename)
    except (OSError, ValueError, ImportError):
        return None
    try:
        if reader.bpp == 8 and not reader.grey_palette:   # indices that are not grey levels: the general path
            return None
        start, end = int(start or 0), min(int(end), reader.frame_count) if end else reader.frame_count
        layout = reader.record_layout(start, end)
        if layout is None:
            return None
        off0, rec, nbytes = layout
        slot = ring.take(nbytes)
        host = slot["host"][:nbytes]
        if device is None:
            device = torch.cuda.current_device()
        dev = torch.device("cuda", int(device))
        torch.cuda.set_device(dev)                       # (this is the decode thread: its current device is its own)
        copy_stream = getattr(_TLS, "decode_stream", None)
        if copy_stream is None or copy_stream.device != dev:
            copy_stream = _TLS.decode_stream = torch.cuda.Stream(dev)
        # (device memory comes back to torch's allocator only when a chunk's results have arrived -- _submit_chunk's
        # keep-alive list -- so a block handed out here is not in use by anything still queued on the GPU)
        staged = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        view = host.numpy()
        spans = [(a, min(nbytes, a + _UPLOAD_PIECE)) for a in range(0, nbytes, _UPLOAD_PIECE)]
        from concurrent.futures import ThreadPoolExecutor
        if not _UPLOAD_IN_DECODER:
            reader.read_region_into(off0, view, threads=_DECODE_THREADS)
            return _RawChunk(slot, nbytes, end - start, rec, reader.height, reader.width, reader.bpp, reader.stride,
                             reader.bottom_up)
        with ThreadPoolExecutor(max_workers=_DECODE_THREADS) as pool:
            reads = [pool.submit(reader.read_region_into, off0 + a, view[a:b], 1) for a, b in spans]
            try:
                with torch.cuda.stream(copy_stream):
                    for (a, b), done in zip(spans, reads):
                        done.result()
                        staged[a:b].copy_(host[a:b], non_blocking=True)
                    uploaded = torch.cuda.Event()
                    uploaded.record(copy_stream)
            except BaseException:
                copy_stream.synchronize()                 # a failed read: no copy may still be reading the buffer
                raise                                     # when the ring hands it to the next driver
        slot["busy"] = uploaded                           # the ring hands this buffer out again only after the copies
        return _RawChunk(slot, nbytes, end - start, rec, reader.height, reader.width, reader.bpp, reader.stride,
                         reader.bottom_up, staged, uploaded)
    finally:
        reader.close()


def _load_chunk_gray_avi(video_filename, start, end):
    """The same short cut one step less direct, for 8-bit AVI files whose frames are NOT evenly spaced (other chunks in
    between): frame by frame from the mapping into one pinned buffer, rows turned over on the way (a few copy threads;
    numpy releases the GIL).  The list handed on holds views of the buffer.  Returns None when it does not apply."""
    if cv2 is not None or not _pristine("load_video_frames"):
        return None
    try:
        reader = _avi.AviReader(video_filename)
    except (OSError, ValueError):
        return None
    try:
        if not reader.grey_palette:                     # colour, or indices that are not grey levels: the general path
            return None
        start, end = int(start or 0), min(int(end), reader.frame_count) if end else reader.frame_count
        n = end - start
        if n <= 0:
            return None
        import torch
        shape = (n, reader.height, reader.width)
        pinned = torch.empty(shape, dtype=torch.uint8, pin_memory=True) if torch.cuda.is_available() else None
        buf = pinned.numpy() if pinned is not None else np.empty(shape, np.uint8)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=_DECODE_THREADS) as pool:
            list(pool.map(lambda k: reader.read_into(start + k, buf[k]), range(n)))
        return list(buf), pinned
    finally:
        reader.close()


def _load_chunk(video_filename, start, end, ring=None, device=None):
    """Decode frames [start, end).  With a staging ring (the asynchronous chunk driver): a _RawChunk when the file allows
    it.  Otherwise (frames, pinned): the frames through the module's load_video_frames -- or the 8-bit AVI short cut --
    and, when they are plain greyscale uint8 frames of one size, the same frames stacked in pinned host memory for the
    asynchronous upload."""
    raw = _load_chunk_raw_avi(video_filename, start, end, ring, device)
    if raw is not None:
        return raw
    fast = _load_chunk_gray_avi(video_filename, start, end)
    if fast is not None:
        return fast
    frames = load_video_frames(video_filename, frames_start=start, frames_end=end)
    pinned = None
    try:
        import torch
        f0 = np.asarray(frames[0]) if len(frames) else None
        if f0 is not None and f0.ndim == 2 and f0.dtype == np.uint8 and torch.cuda.is_available() and \
                all(isinstance(f, np.ndarray) and f.shape == f0.shape and f.dtype == np.uint8 for f in frames):
            pinned = torch.empty((len(frames),) + f0.shape, dtype=torch.uint8, pin_memory=True)
            np.stack(frames, out=pinned.numpy())
    except Exception:                                   # anything unusual: the synchronous path copes or raises
        pinned = None
    return frames, pinned


def _submit_chunk(chunk, idx, kw):
    """Start one chunk; returns a callable that yields detect_damaged_pixels' four arrays (and, from the asynchronous
    path, the chunk's flow metrics as a fifth: the chunk driver reports how close they came to the threshold).  The
    asynchronous path (pinned -> HBM on a side stream, ipx_detect_chunk enqueued, results copied back behind it) is taken
    when nothing has to be plotted and no stage of this module has been replaced; otherwise the chunk goes
    through the module-global detect_damaged_pixels right here."""
    raw = chunk if isinstance(chunk, _RawChunk) else None
    frames, pinned = (None, None) if raw is not None else chunk
    n = len(raw) if raw is not None else len(frames)
    can_async = (raw is not None or pinned is not None) and n >= 2 and not kw["plot"] and _pristine(*_STAGES) and \
        _pristine("detect_damaged_pixels")
    if not can_async:
        res = detect_damaged_pixels(raw.frames() if raw is not None else frames, idx, **kw)
        return lambda: res
    import torch
    settings = _prepare_settings(kw["params"])
    with _Scoped(**_engine_params(settings)) as eng:
        copy_stream = getattr(_TLS, "copy_stream", None)
        if copy_stream is None or copy_stream.device != eng.device:
            copy_stream = _TLS.copy_stream = torch.cuda.Stream(eng.device)
        if raw is not None and raw.staged is not None and raw.staged.device == eng.device:
            staged, uploaded = raw.staged, raw.uploaded   # the decode thread has sent the bytes on already
        else:
            src = raw.slot["host"][:raw.nbytes] if raw is not None else pinned
            staged = torch.empty(src.shape, dtype=torch.uint8, device=eng.device)
            with torch.cuda.stream(copy_stream):
                staged.copy_(src, non_blocking=True)
                uploaded = torch.cuda.Event()
                uploaded.record(copy_stream)
            if raw is not None:
                raw.slot["busy"] = uploaded              # the ring hands this buffer out again only after the copy
        torch.cuda.current_stream(eng.device).wait_event(uploaded)
        if raw is not None:                              # rows the right way up, padding dropped, colour -> grey: on the GPU
            dev = eng.unpack_dib(staged, raw.rec, n, raw.height, raw.width, raw.bpp, raw.stride, raw.bottom_up)
        else:
            dev = staged
        handle = eng.detect_chunk_async(dev)

    def finish():
        r = handle.result()
        del keepalive[:]
        return r["total_counts"], r["cluster_counts"], r["avg_sizes"], r["avg_brightnesses"], r["flows"]
    keepalive = [pinned, staged, dev]
    return finish


def _checkpoint_fingerprint(video_filename, num_frames, step_size, params):
    """What a checkpointed chunk depends on: the video file (name, size, modification time, frame count), the
    chunking, the effective settings and the library version."""
    import json
    from image_processing_amd import _ipx
    try:
        st = os.stat(video_filename)
        ident = [os.path.basename(str(video_filename)), int(st.st_size), int(st.st_mtime_ns)]
    except OSError:
        ident = [os.path.basename(str(video_filename)), -1, -1]
    settings = _prepare_settings(params)
    eff = {k: getattr(settings, k) for k in sorted(_hostmath.DEFAULT_SETTINGS) if k != "number_of_plots"}
    return json.dumps(dict(video=ident, frames=int(num_frames), step_size=int(step_size), settings=eff,
                           ipx_version=int(_ipx.lib.ipx_version())), sort_keys=True, default=float)


def _resolve_devices(gpus):
    """``gpus`` of main() -> list of device ordinals, or None for "this thread's current device".
    None reads the environment variable IPX_GPUS (a count or a comma-separated list of ordinals); a count N means
    devices 0 .. N-1; a list may name a device more than once (two pipelines on one GPU: how a one-GPU box rehearses it)."""
    if gpus is None:
        env = os.environ.get("IPX_GPUS", "").strip()
        if not env:
            return None
        gpus = [int(x) for x in env.split(",") if x.strip()] if "," in env else int(env)
    if isinstance(gpus, (int, np.integer)) and not isinstance(gpus, bool):
        n = int(gpus)
        if n < 1:
            raise ValueError("gpus must be at least 1")
        import torch
        if n > torch.cuda.device_count():
            raise ValueError("gpus=%d but this node shows %d GPU(s)" % (n, torch.cuda.device_count()))
        return list(range(n)) if n > 1 else None
    devs = [int(d) for d in gpus]
    if not devs:
        raise ValueError("gpus: empty device list")
    return devs


def _warm_up(video_filename, n_frames, kw):
    """What the first chunk of a process would otherwise wait for after it has been read: this thread's libipx context
    (30 ms) and the flow stage's plan and arenas.  Done on the driver's thread while the decode thread reads that chunk
    (measured: first delivery of a process after 167 instead of 191 ms; what remains is the pinning of the first staging
    buffer, 55 ms per GB).  Best effort: the chunk's own call makes what is missing."""
    if kw["plot"] or not _pristine(*_STAGES) or not _pristine("detect_damaged_pixels"):
        return
    try:
        if cv2 is None:
            reader = _avi.AviReader(video_filename)
            try:
                H, W = reader.height, reader.width
            finally:
                reader.close()
        else:
            cap, _ = _open_capture(video_filename)
            H, W = int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT)), int(cap.get(cv2.CAP_PROP_FRAME_WIDTH))
            cap.release()
        if H >= 2 and W >= 2 and n_frames >= 2:
            _engine().reserve_flow(n_frames, H, W)
    except Exception:                                   # noqa: BLE001
        pass


def _chunk_pipeline(video_filename, spans, order, kw, deliver, announce):
    """The double-buffered chunk loop on the calling thread's current device: a second thread decodes the next
    chunk of ``order`` (indices into ``spans``) into pinned memory while the current one is on the GPU, and a chunk
    is enqueued before the host turns the previous one's integers into the float series.  ``deliver(idx, res)`` gets
    every finished chunk; ``announce(idx)`` is called when a chunk is taken up."""
    if not order:
        return
    from concurrent.futures import ThreadPoolExecutor
    import time
    timeline = _TIMELINE
    try:
        import torch
        device = torch.cuda.current_device() if torch.cuda.is_available() else None
    except ImportError:
        device = None
    ahead = _prefetch_take(video_filename, spans[order[0]], device)     # announced with queue_next_video()?
    ring = ahead[0] if ahead else _ring_acquire()

    def finish(done, idx, t_taken):
        res = done()
        if timeline is not None:
            timeline.append((idx, t_taken, time.perf_counter()))
        deliver(idx, res)
    try:
        with ThreadPoolExecutor(max_workers=1) as decoder:
            nxt = ahead[1] if ahead else decoder.submit(_load_chunk, video_filename, *spans[order[0]], ring, device)
            if device is not None:
                _warm_up(video_filename, max(b - a for a, b in (spans[i] for i in order)), kw)
            pending = None                               # (completion, idx, time taken up) of the chunk on the GPU
            for k, idx in enumerate(order):
                chunk = nxt.result()
                if k + 1 < len(order):
                    nxt = decoder.submit(_load_chunk, video_filename, *spans[order[k + 1]], ring, device)
                else:
                    _prefetch_start(device)              # nothing left to read here: the next video's first chunk, if announced
                announce(idx)
                t_taken = time.perf_counter()
                done = _submit_chunk(chunk, idx, kw)
                if pending is not None:
                    finish(*pending)
                pending = (done, idx, t_taken)
            if pending is not None:
                finish(*pending)
    finally:
        _ring_release(ring)


def _chunks_on_group(video_filename, spans, order, kw, devices, deliver, announce):
    """Fewer chunks than GPUs: every chunk is sharded by frame range over all ``devices`` inside libipx
    (ipx_group_detect_chunk: one host thread, context and stream per device, halo frames uploaded with each range,
    persistence / static-map exchanges over RCCL or peer copies) instead of leaving GPUs idle."""
    from concurrent.futures import ThreadPoolExecutor
    from image_processing_amd.engine import Group
    settings = _prepare_settings(kw["params"])
    group = Group(devices, params=_engine_params(settings))
    try:
        with ThreadPoolExecutor(max_workers=1) as decoder:
            nxt = decoder.submit(_load_chunk, video_filename, *spans[order[0]])
            pending = None                               # (idx, handle) of the chunk the ranks are computing

            def finish():
                r = group.wait(pending[1])
                deliver(pending[0], (r["total_counts"], r["cluster_counts"], r["avg_sizes"], r["avg_brightnesses"], r["flows"]))
            for k, idx in enumerate(order):
                frames, pinned = nxt.result()
                if k + 1 < len(order):
                    nxt = decoder.submit(_load_chunk, video_filename, *spans[order[k + 1]])
                announce(idx)
                host = pinned.numpy() if pinned is not None else _as_u8(np.stack([np.asarray(f) for f in frames]))
                T = max(int(settings.consecutive_threshold) - 1, 1)
                if len(host) < len(devices) * max(T, 2):         # too short to cut: one device, the ordinary way
                    if pending is not None:
                        finish()
                        pending = None
                    import torch
                    with torch.cuda.device(devices[0]):
                        deliver(idx, _submit_chunk((frames, pinned), idx, kw)())
                    continue
                handle = group.submit(host)              # its upload starts now, under the chunk before
                handle["keepalive"] = pinned
                if pending is not None:
                    finish()
                pending = (idx, handle)
            if pending is not None:
                finish()
    finally:
        group.close()


def main(video_filename, average_time=1.0, max_chunks=None, step_size=1000, plot=False, show_plots=False,
         save_plots=False, output_folder="results", visualise_damaged_pixels_plot=True, plot_damaged_pixels_plot=True,
         plot_heatmap_plot=True, params=None, *, checkpoint_folder=None, gpus=None):
    """ip.py:1136-1247: process a video in chunks of ``step_size`` frames and return the per-window
    averages (``average_time`` seconds each) of the four per-frame series, plus the window mid-times.

    ``checkpoint_folder`` (not in the reference, which keeps everything in RAM until the end): every finished
    chunk's four series are written to ``chunk_<start>_<end>.npz`` there together with a fingerprint of the
    video file, the chunking, the settings and the library version; a chunk whose file exists AND carries this
    run's fingerprint is neither decoded nor processed again (resume after an interruption; no plots for such
    chunks); any other file of that name is recomputed and overwritten.

    ``gpus`` (not in the reference, which is one process on CPU cores): a count or a list of device ordinals;
    default: the environment variable IPX_GPUS, else the calling thread's current device only.  Chunks are
    independent of each other (ip.py:1204-1227: each is loaded and processed on its own), so chunk k goes to device
    k mod N, every device with its own decode thread, libipx context and upload stream; when fewer chunks than
    devices are left to do, each chunk is sharded by frame range over all devices instead (ipx_group_detect_chunk).
    The results are those of a single-device run, bit for bit.  A run that plots, or in which a stage of this module
    has been replaced, stays on one device (those paths run stage by stage on the calling thread)."""
    num_frames, FPS = _video_info(video_filename)
    edges = np.arange(0, num_frames, step_size)
    if edges[-1] != num_frames:
        edges = np.concatenate([edges, [num_frames]])
    if max_chunks is not None:
        edges = edges[:max_chunks + 1]
    step = int(round(FPS * average_time))
    spans = [(int(edges[i]), int(edges[i + 1])) for i in range(len(edges) - 1)]
    kw = dict(plot=plot, show_plots=show_plots, save_plots=save_plots, output_folder=output_folder,
              visualise_damaged_pixels_plot=visualise_damaged_pixels_plot,
              plot_damaged_pixels_plot=plot_damaged_pixels_plot, plot_heatmap_plot=plot_heatmap_plot, params=params)

    def ckpt_path(span):
        return os.path.join(checkpoint_folder, "chunk_%09d_%09d.npz" % span) if checkpoint_folder else None

    fingerprint = _checkpoint_fingerprint(video_filename, num_frames, step_size, params) if checkpoint_folder else ""

    def from_checkpoint(span):
        """The chunk's four series if its file was written for THIS video, chunking and parameter set by this
        library version; anything else (missing, other fingerprint, unreadable) means: compute it."""
        path = ckpt_path(span)
        if not path or not os.path.exists(path):
            return None
        try:
            with np.load(path) as z:
                if str(z["fingerprint"]) != fingerprint or int(z["n_frames"]) != span[1] - span[0]:
                    return None
                return tuple(z[k] for k in ("total_counts", "cluster_counts", "avg_sizes", "avg_brightnesses"))
        except Exception:                               # noqa: BLE001 - truncated / foreign file: recompute, overwrite
            return None

    results = [from_checkpoint(sp) for sp in spans]     # per chunk: the four series
    todo = [i for i in range(len(spans)) if results[i] is None]
    for i, r in enumerate(results):
        if r is not None:
            print(f"chunk {i}: frames {spans[i][0]}–{spans[i][1]} taken from {ckpt_path(spans[i])}")
    lock = _threading.Lock()

    flow_threshold = float(_prepare_settings(params).flow_threshold)
    margins = []                                         # per chunk computed here on the fused path: hostmath.flow_margin

    def deliver(idx, res):
        """A chunk computed in this run: keep it and (over)write its checkpoint."""
        if len(res) > 4:
            with lock:
                margins.append(_hostmath.flow_margin(res[4], flow_threshold, rel=1e-4))
        res = tuple(res[:4])
        results[idx] = res
        path = ckpt_path(spans[idx])
        if path:
            with lock:
                os.makedirs(checkpoint_folder, exist_ok=True)
            tmp = path + ".tmp.npz"
            np.savez(tmp, total_counts=res[0], cluster_counts=res[1], avg_sizes=res[2], avg_brightnesses=res[3],
                     n_frames=spans[idx][1] - spans[idx][0], fingerprint=fingerprint)
            os.replace(tmp, path)

    def announce(idx):
        print(f"processing chunk {idx}: frames {spans[idx][0]}–{spans[idx][1]}")

    devices = _resolve_devices(gpus)
    if devices is not None and len(devices) > 1 and (plot or not _pristine(*_STAGES) or
                                                     not _pristine("detect_damaged_pixels", "load_video_frames")):
        devices = devices[:1]
    if devices is None:
        _chunk_pipeline(video_filename, spans, todo, kw, deliver, announce)
    elif len(devices) == 1:
        import torch
        with torch.cuda.device(devices[0]):
            _chunk_pipeline(video_filename, spans, todo, kw, deliver, announce)
    elif todo and len(todo) < len(devices):
        _chunks_on_group(video_filename, spans, todo, kw, devices, deliver, announce)
    elif todo:
        from concurrent.futures import ThreadPoolExecutor
        import torch

        def on_device(slot):
            torch.cuda.set_device(devices[slot])         # the device of THIS thread; its libipx context is thread-local
            _chunk_pipeline(video_filename, spans, todo[slot::len(devices)], kw, deliver, announce)
        with ThreadPoolExecutor(max_workers=len(devices)) as pool:
            for f in [pool.submit(on_device, s) for s in range(len(devices))]:
                f.result()
    counts, clusters, sizes, brightness = (np.concatenate([r[k] for r in results]) for k in range(4))
    averages_counts = chunked_nanmean(counts, step)
    times = ((np.arange(len(averages_counts)) * step) + step / 2) / FPS
    # The discard list (ip.py:559) is the one place where the float metric of the motion test decides an integer
    # result, and the Farneback numerics behind it are restated from OpenCV's published algorithm, not pinned by a
    # reference fixture: say so when a frame comes close enough to the threshold for the last bits to matter.
    near = sum(m["frames_within_rel"] for m in margins)
    if near:
        print(f"warning: {near} frame(s) have a motion metric within 1e-4 (relative) of flow_threshold={flow_threshold}; "
              f"whether they are kept depends on the last bits of the optical flow")
    return {
        "averages_counts": averages_counts,
        "averages_clusters": chunked_nanmean(clusters, step),
        "averages_size": chunked_nanmean(sizes, step),
        "averages_brightness": chunked_nanmean(brightness, step),
        "times": times,
        # (not in the reference) min |metric - flow_threshold| over the frames computed in this run; None when no chunk
        # went through the fused path (all from checkpoints, or a replaced stage / plotting run)
        "min_flow_margin": min((m["min_margin"] for m in margins), default=None),
    }


for _n in _STAGES:
    _ORIG[_n] = globals()[_n]
del _n
_ORIG["detect_damaged_pixels"] = detect_damaged_pixels
_ORIG["load_video_frames"] = load_video_frames


def _cli_options(argv=None):
    """Parse the reference's command line (ip.py:1250-1316: same flags, every value a string) into
    dict(video_filename, main_kwargs, show_plots, save_plots, average_plots)."""
    import argparse
    parser = argparse.ArgumentParser()
    parser.add_argument("-vf", "--video_filename", help="Path to the AVI file")
    parser.add_argument("-at", "--average_time", help="How many seconds to average over in the final summaries")
    parser.add_argument("-mc", "--max_chunks", help="If not None, only process that many chunks (for quick tests)")
    parser.add_argument("-ss", "--step_size", help="The number of frames in each chunk")
    parser.add_argument("-shp", "--show_plots", help="(Default=False) True or False: show plots")
    parser.add_argument("-svp", "--save_plots", help="(Default=False) True or False: save plots")
    parser.add_argument("-vdpp", "--visualise_damaged_pixels_plot", help="(Default=False) True or False")
    parser.add_argument("-pdpp", "--plot_damaged_pixels_plot", help="(Default=True) True or False")
    parser.add_argument("-php", "--plot_heatmap_plot", help="(Default=True) True or False")
    parser.add_argument("-cp", "--chunk_plots", help="(Default=True) True or False: per-chunk plots")
    parser.add_argument("-ap", "--average_plots", help="(Default=True) True or False: final averages plot")
    parser.add_argument("-of", "--output_folder", help="Location of where saved plots are saved")
    parser.add_argument("-gp", "--gpus", help="(not in the reference) number of GPUs or comma-separated device ordinals; chunks are dealt over them")
    parser.add_argument("-ckf", "--checkpoint_folder", help="(not in the reference) per-chunk .npz results; finished chunks are skipped on a rerun")
    for short, name in (("-ct", "consecutive_threshold"), ("-bt", "brightness_threshold"), ("-ft", "flow_threshold"),
                        ("-st", "static_threshold"), ("-mncs", "min_cluster_size"), ("-mxcs", "max_cluster_size"),
                        ("-mnc", "min_circularity"), ("-swr", "sliding_window_radius"), ("-np", "number_of_plots")):
        parser.add_argument(short, "--" + name)
    args = parser.parse_args(argv)

    def flag(v, default):
        return (v.lower() == "true") if v else default

    def num(v, cast, default):
        return cast(v) if v else default

    if args.max_chunks:
        max_chunks = None if args.max_chunks.lower() == "none" else int(args.max_chunks)
    else:
        max_chunks = 2                                                  # ip.py:1293
    show_plots, save_plots = flag(args.show_plots, False), flag(args.save_plots, False)
    plot = (show_plots or save_plots) and flag(args.chunk_plots, True)
    output_folder = args.output_folder if args.output_folder else "results"
    params = {
        "consecutive_threshold": num(args.consecutive_threshold, int, 2),
        "brightness_threshold": num(args.brightness_threshold, int, 170),
        "flow_threshold": num(args.flow_threshold, float, 2.0),
        "static_threshold": num(args.static_threshold, int, 50),
        "min_cluster_size": num(args.min_cluster_size, int, 5),
        "max_cluster_size": num(args.max_cluster_size, int, 20),
        "min_circularity": num(args.min_circularity, float, 0.1),
        "sliding_window_radius": num(args.sliding_window_radius, int, 3),
        "number_of_plots": num(args.number_of_plots, int, 20),
    }
    main_kwargs = dict(
        average_time=num(args.average_time, float, 1.0), max_chunks=max_chunks,
        step_size=num(args.step_size, int, 1000), plot=plot, show_plots=show_plots, save_plots=save_plots,
        output_folder=output_folder,
        visualise_damaged_pixels_plot=flag(args.visualise_damaged_pixels_plot, False),
        plot_damaged_pixels_plot=flag(args.plot_damaged_pixels_plot, True),
        plot_heatmap_plot=flag(args.plot_heatmap_plot, True), params=params,
        checkpoint_folder=args.checkpoint_folder or None,
        gpus=([int(x) for x in args.gpus.split(",")] if "," in args.gpus else int(args.gpus)) if args.gpus else None)
    return dict(video_filename=args.video_filename if args.video_filename else "11_01_H_170726081325.avi",
                main_kwargs=main_kwargs, show_plots=show_plots, save_plots=save_plots,
                average_plots=flag(args.average_plots, True))


def _cli(argv=None):
    """The reference's command line (ip.py:1250-1350)."""
    opts = _cli_options(argv)
    show_plots, save_plots, output_folder = opts["show_plots"], opts["save_plots"], opts["main_kwargs"]["output_folder"]
    results = main(opts["video_filename"], **opts["main_kwargs"])
    if opts["average_plots"] and HAS_MATPLOTLIB:
        plot_all_chunks(results["averages_counts"], results["averages_clusters"], results["averages_size"],
                        results["averages_brightness"], results["times"], show_plots=show_plots,
                        save_plots=save_plots, output_folder=output_folder)
    else:
        print("counts:", results["averages_counts"])
        print("clusters:", results["averages_clusters"])
        print("sizes:", results["averages_size"])
        print("brightness:", results["averages_brightness"])
        print("times:", results["times"])
    print("\nDone")
    return results


if __name__ == "__main__":
    _cli()

"""CPU restatement of the per-frame damaged-pixel detection stages.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c).  Each function cites the lines of
/root/reference/image_processing.py ("ip.py") whose behaviour it restates; the
third-party arithmetic (cv2 / numba) follows SURVEY.md App. A and lives in
oracle.c.  Discarded frames are ``None`` entries of a mask list, exactly as in
ip.py:557.
"""
import math

import numpy as np

from . import cref

DEFAULTS = dict(  # ip.py:167-177
    consecutive_threshold=2, brightness_threshold=170, flow_threshold=2.0,
    static_threshold=50, min_cluster_size=5, max_cluster_size=20,
    min_circularity=0.1, sliding_window_radius=3, number_of_plots=20,
)


def settings(params=None):
    """ip.py:154-180 -- a plain dict here; the values are what matters."""
    d = dict(DEFAULTS)
    if params:
        d.update(params)
    return d


# -- background ------------------------------------------------------------------------------
def background_window(n_frames, index, radius):
    """Indices of the temporal neighbours of ``index`` (centre excluded), ip.py:189-198."""
    lo, hi = max(0, index - radius), min(n_frames, index + radius + 1)
    return [k for k in range(lo, hi) if k != index]


def compute_background(frames, index, radius, coeff=1.0):
    """ip.py:183-200 + 203-252: f32 neighbours -> outlier-trimmed temporal mean (f32)."""
    idx = background_window(len(frames), index, radius)
    stack = np.asarray([frames[k] for k in idx], dtype=np.float32)
    return cref.find_background_f32(stack, coeff)


# -- raw mask --------------------------------------------------------------------------------
def damaged_threshold(background):
    """ip.py:638 -- evaluated in the dtype of ``background`` (f32 in the pipeline)."""
    return np.maximum(30, 30 + (background / 255) * (255 - 30))


def local_mean_30(frame):
    """ip.py:643-644: cv2.blur(frame as f64, (30,30))."""
    return cref.blur_f64(np.asarray(frame, dtype=np.float64), (30, 30))


def get_damaged_pixel_mask(frame, background):
    """ip.py:635-648 -> (u8 mask, threshold array)."""
    thr = damaged_threshold(background)
    hit = frame > thr
    hit &= frame > local_mean_30(frame)
    return hit.astype(np.uint8), thr


def raw_damaged_mask(frame, background):
    """ip.py:255-267."""
    return get_damaged_pixel_mask(frame, background)[0].astype(bool)


# -- connected components --------------------------------------------------------------------
def connected_components(mask):
    """cv2.connectedComponentsWithStats(mask, connectivity=8): (num_labels, labels, areas)."""
    n, labels = cref.ccl8(mask)
    areas = np.bincount(labels.ravel(), minlength=n)
    return n, labels, areas


def remove_bright_regions(background, brightness_threshold, damaged, max_cluster_size):
    """ip.py:731-767: drop damaged components that touch a large bright-background blob."""
    damaged = np.asarray(damaged).astype(bool)
    bright = background > brightness_threshold
    _, lab_b, area_b = connected_components(bright)
    big = area_b >= max_cluster_size
    big[0] = False
    remove = big[lab_b]
    n_d, lab_d, _ = connected_components(damaged)
    tainted = np.zeros(n_d, bool)
    tainted[np.unique(lab_d[remove & damaged])] = True
    tainted[0] = True
    return ~tainted[lab_d]


# -- temporal maps ---------------------------------------------------------------------------
def compute_persistent_mask(masks, consecutive_threshold):
    """ip.py:270-289: longest run of consecutive flagged frames >= T; None resets runs."""
    shape = masks[0].shape
    run = np.zeros(shape, np.int32)
    longest = np.zeros(shape, np.int32)
    for m in masks:
        if m is None:
            run[...] = 0
        else:
            run = np.where(m, run + 1, 0).astype(np.int32)
        np.maximum(longest, run, out=longest)
    return longest >= consecutive_threshold


def filter_consecutive_pixels(masks, persistent):
    """ip.py:292-310."""
    out, counts = [], []
    keep = ~persistent
    for m in masks:
        if m is None:
            out.append(None)
            counts.append(np.nan)
        else:
            f = m & keep
            out.append(f)
            counts.append(int(f.sum()))
    return out, counts


def compute_static_mask(frames, masks, brightness_threshold, static_threshold, min_valid_frames):
    """ip.py:438-465: heat / valid percentage map > static_threshold."""
    shape = np.asarray(frames[0]).shape
    heat = np.zeros(shape, np.uint64)
    valid = np.zeros(shape, np.int64)
    for f, m in zip(frames, masks):
        mm = np.zeros(shape, bool) if m is None else np.asarray(m).astype(bool)
        heat += mm.astype(np.uint64)
        valid += ~((np.asarray(f) > brightness_threshold) & ~mm)
    pct = np.zeros(shape, float)
    good = valid > min_valid_frames
    pct[good] = (heat[good] / valid[good]) * 100
    return pct > static_threshold


def apply_static_suppression(masks, persistent, static_mask):
    """ip.py:468-481."""
    keep = ~(persistent | static_mask)
    out, counts = [], []
    for m in masks:
        if m is None:
            out.append(None)
            counts.append(np.nan)
        else:
            f = m & keep
            out.append(f)
            counts.append(int(f.sum()))
    return out, np.array(counts, dtype=float)


# -- bright-area estimates -------------------------------------------------------------------
def estimate_from_counts(n_mask, n_low, n_high):
    """ip.py:807-816 on the three integer counters; Python round = half-to-even."""
    if n_low > 0:
        density = np.int64(n_mask) / np.int64(n_low)
        return round(density * np.int64(n_high))
    return np.nan


def estimate_damaged_pixels_in_bright_areas(frames, masks, brightness_threshold=170):
    """ip.py:770-821."""
    out = np.full(len(frames), np.nan, np.float64)
    for i, f in enumerate(frames):
        f = np.asarray(f)
        m = np.zeros(f.shape, bool) if masks[i] is None else np.asarray(masks[i]).astype(bool)
        low = int(((f < brightness_threshold) & ~m).sum())
        high = int(((f >= brightness_threshold) & ~m).sum())
        out[i] = estimate_from_counts(int(m.sum()), low, high)
    return out


def find_bright_area_estimates(frames, masks, brightness_threshold):
    """ip.py:824-847 -- note the inner estimate always uses the default 170 (ip.py:836)."""
    out = np.full(len(frames), np.nan, np.float64)
    est = estimate_damaged_pixels_in_bright_areas(frames, masks)
    for i, (f, m) in enumerate(zip(frames, masks)):
        if m is not None and ((np.asarray(f) > brightness_threshold) & ~m).sum() > 0:
            out[i] = est[i]
    return out


def get_final_count(masks, bright_estimates):
    """ip.py:351-365."""
    return np.array([(np.sum(m) if m is not None else np.nan) + (e if not np.isnan(e) else 0)
                     for m, e in zip(masks, bright_estimates)], dtype=float)


# -- cluster statistics ----------------------------------------------------------------------
def morph_close_cross3(mask):
    """cv2.morphologyEx(mask, MORPH_CLOSE, ellipse 3x3 = cross), SURVEY App. A.4:
    dilate (outside = 0) then erode (outside ignored)."""
    m = np.asarray(mask).astype(bool)

    def shifts(a, fill):
        up = np.full_like(a, fill); up[1:] = a[:-1]
        dn = np.full_like(a, fill); dn[:-1] = a[1:]
        lf = np.full_like(a, fill); lf[:, 1:] = a[:, :-1]
        rt = np.full_like(a, fill); rt[:, :-1] = a[:, 1:]
        return up, dn, lf, rt

    u, d, l, r = shifts(m, False)
    dil = m | u | d | l | r
    u, d, l, r = shifts(dil, True)
    return dil & u & d & l & r


SQRT2_F32 = float(np.sqrt(np.float32(2.0)))  # arcLength: float sqrt(dx*dx+dy*dy) per segment


def contour_perimeter(component_img, x0, y0):
    """cv2.arcLength(findContours(RETR_EXTERNAL, CHAIN_APPROX_NONE)[0], True), ip.py:691-693."""
    n, na, nd = cref.outer_contour(component_img, x0, y0)
    if n <= 1:
        return 0.0
    return na * 1.0 + nd * SQRT2_F32


def filter_damaged_pixel_clusters(frame, mask, min_cluster_size, max_cluster_size, min_circularity,
                                  circularity_size_threshold=10):
    """ip.py:651-728 -> (cleaned mask, count, mean area, mean brightness)."""
    frame = np.asarray(frame)
    closed = morph_close_cross3(mask)
    n, labels, areas_all = connected_components(closed)
    cleaned = np.zeros(closed.shape, bool)
    areas = areas_all[1:]
    valid = np.flatnonzero((areas >= min_cluster_size) & (areas <= max_cluster_size)) + 1
    if len(valid) == 0:
        return cleaned, 0, 0.0, float("nan")
    circ = np.zeros(n, np.float32)
    for lab in valid:
        if areas_all[lab] >= circularity_size_threshold:
            comp = labels == lab
            ys, xs = np.nonzero(comp)          # raster order: first entry = first pixel
            per = contour_perimeter(comp, xs[0], ys[0])
            if per > 0:
                circ[lab] = 4 * np.pi * (np.int32(areas_all[lab]) / (per ** 2))
    kept = [lab for lab in valid
            if areas_all[lab] < circularity_size_threshold or circ[lab] >= min_circularity]
    if not kept:
        return cleaned, 0, 0.0, float("nan")
    cleaned = np.isin(labels, kept)
    sums = np.bincount(labels.ravel(), weights=frame.ravel().astype(np.float32), minlength=n)
    used = np.array(kept)
    a, b = areas_all[used], sums[used]
    return cleaned, len(kept), float(np.mean(a)), float(np.sum(b) / np.sum(a))


def compute_cluster_stats(frames, masks, flows, st):
    """ip.py:332-348."""
    res = []
    for f, m, fl in zip(frames, masks, flows):
        if m is None or fl > st["flow_threshold"]:
            res.append((0.0, 0.0, 0.0))
        else:
            _, c, s, b = filter_damaged_pixel_clusters(f, m, st["min_cluster_size"],
                                                       st["max_cluster_size"], st["min_circularity"])
            res.append((c, s, b))
    return np.array(res).T


# -- optical flow ----------------------------------------------------------------------------
def flow_between(prev, curr):
    """ip.py:855-868: mean Farneback flow magnitude (np.mean of an f32 array -> f32)."""
    flow = cref.farneback(np.asarray(prev, np.float32), np.asarray(curr, np.float32),
                          0.5, 3, 15, 3, 5, 2.0)
    return np.mean(cref.flow_magnitude(flow))


def compute_optical_flow_metric(frames):
    """ip.py:852-873: flows[0] = 0, flows[i] = metric(frames[i-1], frames[i])."""
    flows = np.zeros(len(frames))
    for i in range(1, len(frames)):
        flows[i] = flow_between(frames[i - 1], frames[i])
    return flows


# -- chunk driver helpers (ip.py:1068-1085) ---------------------------------------------------
def chunked_nanmean(array, step):
    if len(array) == 0:
        return [np.nan]
    array = np.asarray(array, dtype=float)
    n_full = (len(array) // step) * step
    with np.errstate(all="ignore"):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            means = np.nanmean(array[:n_full].reshape(-1, step), axis=1)
            if len(array) % step:
                means = np.append(means, np.nanmean(array[n_full:]))
    return means.tolist()


def isnan(x):
    return isinstance(x, float) and math.isnan(x)

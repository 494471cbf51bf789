"""Host-side float glue of the detection path: what ip.py does with the per-frame integers.

libipx returns integer counters per frame (include/ipx.h, ipx_detect_chunk); the few float
operations the reference applies to them (ip.py:351-365, 807-816, 838-845, 720-726) are a
handful of scalar divisions per frame and stay in Python so that they are evaluated by the
same interpreter arithmetic as the reference (Python ``round`` = round-half-even, float ``**``).
"""
import math

import numpy as np

SQRT2_F32 = float(np.sqrt(np.float32(2.0)))   # cv2.arcLength: float sqrt(dx*dx+dy*dy) per diagonal move

#: reference defaults, ip.py:167-177
DEFAULT_SETTINGS = dict(
    consecutive_threshold=2, brightness_threshold=170, flow_threshold=2.0, static_threshold=50,
    min_cluster_size=5, max_cluster_size=20, min_circularity=0.1, sliding_window_radius=3,
    number_of_plots=20,
)


def pow2_table(max_cluster_size):
    """perimeter ** 2 for every (axis moves, diagonal moves) pair a component of at most
    ``max_cluster_size`` pixels can produce, evaluated by Python's float pow exactly as
    ip.py:695 does (libm pow(p, 2) is not always p*p)."""
    tdim = int(min(max(2 * int(max_cluster_size) + 8, 16), 600))
    a = np.arange(tdim, dtype=np.float64)
    per = a[:, None] * 1.0 + a[None, :] * SQRT2_F32
    flat = per.ravel().tolist()
    out = np.array([p ** 2 for p in flat], dtype=np.float64)
    return out, tdim


def bright_area_estimate(n_mask, n_mask_low, n_mask_gt, n_low, n_gt, n_pixels):
    """ip.py:800-816 + 838-845 from the integer counters of one kept frame.

    n_mask = sum(final); n_mask_low = #(final & frame < 170); n_mask_gt = #(final & frame > bt);
    n_low = #(frame < 170); n_gt = #(frame > bt); n_pixels = H*W.  Returns the estimate or nan.
    """
    if n_gt - n_mask_gt <= 0:                       # ip.py:840-845: no bright pixel outside the mask
        return float("nan")
    low = n_low - n_mask_low                        # (frame < 170) & ~mask
    high = (n_pixels - n_low) - (n_mask - n_mask_low)   # (frame >= 170) & ~mask
    if low > 0:
        density = n_mask / low                      # ip.py:809
        return round(density * high)                # ip.py:812-813, Python round (half to even)
    return float("nan")


def chunk_results(keep, counts, cluster, n_pixels):
    """The four float64 arrays of ip.py:632 from ipx_detect_chunk's outputs.

    keep: (n,) u8; counts: (n,5) int64; cluster: (n,3) int64 = {count, sum area, sum brightness}.
    """
    n = len(keep)
    total = np.empty(n, np.float64)
    ccount = np.zeros(n, np.float64)
    csize = np.zeros(n, np.float64)
    cbright = np.zeros(n, np.float64)
    counts = np.asarray(counts).tolist()
    cluster = np.asarray(cluster).tolist()
    for i in range(n):
        if not keep[i]:                             # discarded: nan count (ip.py:360), zeros (ip.py:335)
            total[i] = np.nan
            continue
        m, m_low, m_gt, n_low, n_gt = counts[i]
        est = bright_area_estimate(m, m_low, m_gt, n_low, n_gt, n_pixels)
        total[i] = m + (0 if isinstance(est, float) and math.isnan(est) else est)   # ip.py:359-362
        c, sa, sb = cluster[i]
        if c > 0:
            ccount[i] = c
            csize[i] = sa / c                       # float(np.mean(areas)), ip.py:722
            cbright[i] = sb / sa                    # sum(brightness) / sum(areas), ip.py:723
        else:
            cbright[i] = np.nan                     # ip.py:680 / 707
    return total, ccount, csize, cbright


def flow_margin(flows, flow_threshold, rel=1e-3):
    """How close the motion test comes to its threshold over a chunk (ip.py:559: flow <= flow_threshold keeps a
    frame): min |flow - threshold| over frames 1.. (frame 0's metric is the constant 0), and the number of frames
    within ``rel`` relative of the threshold.  The quantitative cover for a discard list that rests on float
    metrics: a margin far above the metric's tolerance means no frame can flip."""
    f = np.asarray(flows, dtype=np.float64)[1:]
    f = f[np.isfinite(f)]
    if f.size == 0:
        return dict(min_margin=float("inf"), frames_within_rel=0, rel=rel)
    d = np.abs(f - float(flow_threshold))
    return dict(min_margin=float(d.min()), frames_within_rel=int((d <= rel * abs(float(flow_threshold))).sum()), rel=rel)

"""Chunk orchestration of the oracle: ip.py:526-632 (detect_damaged_pixels) restated.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c).
"""
import numpy as np

from . import stages as S


def detect_damaged_pixels(frames, params=None, flows=None, return_intermediates=False, timings=None):
    """(N,H,W) u8 frames -> (total_counts, cluster_counts, avg_sizes, avg_brightnesses),
    four float64 arrays of length N; order of operations is ip.py:550-613.

    ``flows`` may be supplied to bypass the optical-flow metric (used by tests that
    pin the non-flow stages independently of the parity-unpinned Farneback restatement).
    ``timings``: a dict that receives the wall-clock seconds per stage (bench.py's cpu_baseline).
    """
    import time
    tm = timings if timings is not None else {}
    clock = [time.perf_counter()]

    def lap(stage):
        now = time.perf_counter()
        tm[stage] = tm.get(stage, 0.0) + now - clock[0]
        clock[0] = now

    st = S.settings(params)
    frames = np.asarray(frames)
    n = len(frames)
    if flows is None:
        flows = S.compute_optical_flow_metric(frames)           # ip.py:554
    flows = np.asarray(flows, dtype=np.float64)
    lap("optical_flow")
    raw = [None] * n                                            # ip.py:557
    for i in range(n):                                          # ip.py:558-572
        if flows[i] <= st["flow_threshold"]:
            bg = S.compute_background(frames, i, st["sliding_window_radius"])
            lap("background")
            m = S.raw_damaged_mask(frames[i], bg)
            lap("raw_mask")
            raw[i] = S.remove_bright_regions(bg, st["brightness_threshold"], m, st["max_cluster_size"])
            lap("remove_bright_regions")
    persistent = S.compute_persistent_mask(raw, st["consecutive_threshold"])      # ip.py:575
    clean, _ = S.filter_consecutive_pixels(raw, persistent)                         # ip.py:580
    lap("persistence")
    static = S.compute_static_mask(frames, clean, st["brightness_threshold"],
                                   st["static_threshold"], 10)                     # ip.py:584-590
    final, _ = S.apply_static_suppression(clean, persistent, static)                # ip.py:592
    lap("static_map")
    est = S.find_bright_area_estimates(frames, final, st["brightness_threshold"])   # ip.py:599-603
    total = S.get_final_count(final, est)                                           # ip.py:605
    lap("estimates_counts")
    cc, sz, br = S.compute_cluster_stats(frames, final, flows, st)                  # ip.py:608
    lap("cluster_stats")
    if return_intermediates:
        return (total, cc, sz, br), dict(flows=flows, raw=raw, persistent=persistent, clean=clean,
                                         static=static, final=final, estimates=est)
    return total, cc, sz, br

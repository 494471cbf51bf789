"""Generate the golden vectors under tests/golden/ (run in the BUILD container only).

What this does: imports the reference module from /root/reference with two stand-in
modules registered first -- ``numba`` (njit = identity, prange = range) and ``cv2``
(the handful of functions the module calls, each bound to OUR CPU restatement in
oracle/) -- then runs the reference's own Python on small seeded inputs and stores
inputs + outputs as .npz.  Consequences, stated plainly:

* the numpy-only stages (persistence, static map, suppression, estimates, final counts,
  chunk averages) and ALL orchestration semantics (None handling, NaN rules, ordering,
  dtype promotion) are pinned by the reference's own code;
* the cv2 / numba numerics inside those runs are our restatement:
  they are pinned only where the reference's own tests hold a known answer
  (tests/test_oracle_known_answers.py); Farneback, contour perimeter and cvtColor
  remain "parity unpinned".

The reference source never leaves /root/reference; only data is written here.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import cref, stages as S  # noqa: E402


def install_stubs():
    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.njit, nb.prange = njit, range
    sys.modules["numba"] = nb

    cv = types.ModuleType("cv2")
    cv.CAP_PROP_POS_FRAMES, cv.CAP_PROP_FRAME_COUNT, cv.CAP_PROP_FPS = 1, 7, 5
    cv.COLOR_BGR2GRAY, cv.COLOR_GRAY2BGR, cv.COLOR_BGR2RGB = 6, 8, 4
    cv.MORPH_ELLIPSE, cv.MORPH_CLOSE, cv.CC_STAT_AREA = 2, 3, 4
    cv.RETR_EXTERNAL, cv.CHAIN_APPROX_NONE = 0, 1

    cv.blur = lambda src, ksize: cref.blur_f64(src, ksize)

    def ccws(img, connectivity=8):
        n, labels, areas = S.connected_components(img)
        stats = np.zeros((n, 5), np.int32)
        stats[:, 4] = areas
        return n, labels, stats, None
    cv.connectedComponentsWithStats = ccws
    cv.getStructuringElement = lambda shape, ksize: np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    cv.morphologyEx = lambda img, op, kernel: S.morph_close_cross3(img).astype(np.uint8)

    def find_contours(img, mode, method):
        ys, xs = np.nonzero(img)
        if len(ys) == 0:
            return [], None
        pts, _, _ = cref.outer_contour(img, xs[0], ys[0], want_points=True)
        return [pts.reshape(-1, 1, 2)], None
    cv.findContours = find_contours

    def arc_length(contour, closed):
        p = contour.reshape(-1, 2).astype(np.float32)
        if len(p) <= 1:
            return 0.0
        prev = p[-1] if closed else p[0]
        per = 0.0
        for q in p:
            d = q - prev
            per += float(np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1])))
            prev = q
        return per
    cv.arcLength = arc_length

    def farneback(prev, nxt, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
        assert flags == 0
        return cref.farneback(prev, nxt, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma)
    cv.calcOpticalFlowFarneback = farneback
    cv.cartToPolar = lambda x, y: (np.sqrt(x * x + y * y), None)
    cv.cvtColor = lambda img, code: cref.bgr2gray_u8(img)
    sys.modules["cv2"] = cv


def synth_chunk(seed, n, h, w, hot=6.0, stuck=3, bright_block=True, motion_at=()):
    """Small seeded chunk: textured static scene + sensor noise + transient hot pixels,
    a few stuck pixels, one bright block, optional scene shifts (motion) at given frames."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    frames = np.empty((n, h, w), np.uint8)
    stuck_px = [(int(rng.integers(h)), int(rng.integers(w))) for _ in range(stuck)]
    off = 0.0
    for i in range(n):
        if i in motion_at:
            off += 6.0
        x = xx - off
        base = 70 + 30 * np.sin(x / 5.0) * np.cos(yy / 4.0) + 20 * np.sin((x + yy) / 7.0)
        img = base + rng.normal(0, 2.0, (h, w))
        if bright_block:
            img[h // 2:h // 2 + 9, w // 3:w // 3 + 11] = 200
        k = rng.poisson(hot)
        for _ in range(k):
            y, x0 = int(rng.integers(h)), int(rng.integers(w))
            v = int(rng.integers(150, 256))
            img[y, x0] = v
            if rng.random() < 0.3:
                for _ in range(int(rng.integers(1, 12))):
                    yy2 = min(h - 1, max(0, y + int(rng.integers(-2, 3))))
                    xx2 = min(w - 1, max(0, x0 + int(rng.integers(-2, 3))))
                    img[yy2, xx2] = v
        for (y, x0) in stuck_px:
            img[y, x0] = 250
        frames[i] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    return frames


def masks_to_arrays(masks, shape):
    """list of bool arrays / None -> (kept flags u8, stacked bool array)."""
    kept = np.array([m is not None for m in masks], np.uint8)
    st = np.stack([np.zeros(shape, bool) if m is None else np.asarray(m, bool) for m in masks])
    return kept, st


def main():
    install_stubs()
    sys.path.insert(0, "/root/reference")
    import image_processing as ref  # the reference's own module, unmodified

    out = {}
    rng = np.random.default_rng(7)

    # ---- numpy-only stages: pure reference outputs ----
    h, w, n = 9, 13, 14
    raw = [(rng.random((h, w)) < 0.25) for _ in range(n)]
    raw[4] = None
    raw[9] = None
    frames = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
    kept, raw_arr = masks_to_arrays(raw, (h, w))
    for T in (1, 2, 3, 5):
        out[f"persist_T{T}"] = ref.compute_persistent_mask(raw, T)
    pers = ref.compute_persistent_mask(raw, 3)
    clean, ccounts = ref.filter_consecutive_pixels(raw, pers)
    out["tmp_frames"], out["tmp_kept"], out["tmp_raw"] = frames, kept, raw_arr
    out["tmp_clean"] = masks_to_arrays(clean, (h, w))[1]
    out["tmp_clean_counts"] = np.array(ccounts, float)
    for mv in (10, 3):
        for stt in (50, 20, 5):
            out[f"static_mv{mv}_st{stt}"] = ref.compute_static_mask(frames, clean, 170, stt, mv)
    static = ref.compute_static_mask(frames, clean, 170, 5, 3)
    final, fcounts = ref.apply_static_suppression(clean, pers, static)
    out["tmp_final"] = masks_to_arrays(final, (h, w))[1]
    out["tmp_final_counts"] = fcounts
    est = ref.estimate_damaged_pixels_in_bright_areas(frames.astype(np.float64), final)
    out["tmp_estimate"] = est
    for bt in (170, 100, 254):
        out[f"tmp_bright_est_bt{bt}"] = ref.find_bright_area_estimates(frames.astype(np.float64), final, bt)
    out["tmp_final_count"] = ref._get_final_count(final, ref.find_bright_area_estimates(
        frames.astype(np.float64), final, 170))
    arr = np.array([1.0, 2.0, np.nan, 4.0, 5.0, np.nan, np.nan, 8.0])
    out["nanmean_in"] = arr
    for step in (2, 3, 8, 9):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[f"nanmean_step{step}"] = np.array(ref.chunked_nanmean(arr, step))

    # ---- background: reference body run as plain Python (f32 accumulators, SURVEY 8c caveat) ----
    bgf = synth_chunk(11, 9, 12, 16, hot=3.0)
    out["bg_frames"] = bgf
    for r in (1, 3):
        out[f"bg_r{r}"] = np.stack([ref.compute_background(bgf, i, r) for i in range(len(bgf))])
    toy = np.array([[[1, 5], [5, 1]], [[2, 6], [6, 2]], [[200, 200], [200, 200]],
                    [[7, 5], [3, 8]], [[3, 1], [4, 5]]], dtype=float)
    out["bg_toy_r1"] = np.stack([ref.compute_background(toy, i, 1) for i in range(5)])

    # ---- hybrid: reference code over our cv2 restatements ----
    ch = synth_chunk(3, 16, 40, 56, hot=5.0, motion_at=(6, 7, 12))
    out["hy_frames"] = ch
    flows = ref.compute_optical_flow_metric(ch)
    out["hy_flows"] = flows
    bgs, rmasks, cleaned = [], [], []
    for i in range(len(ch)):
        bg = ref.compute_background(ch, i, 3)
        m = ref._raw_damaged_mask(ch[i], bg)
        c = ref.remove_bright_regions(bg, 170, m, 20)
        bgs.append(bg); rmasks.append(m); cleaned.append(c)
    out["hy_bg"], out["hy_rawmask"], out["hy_brightfiltered"] = np.stack(bgs), np.stack(rmasks), np.stack(cleaned)
    res = ref.detect_damaged_pixels(list(ch), 0)
    out["hy_detect"] = np.stack(res)
    res2 = ref.detect_damaged_pixels(list(ch), 0, params=dict(
        consecutive_threshold=3, brightness_threshold=150, flow_threshold=0.05, static_threshold=20,
        min_cluster_size=2, max_cluster_size=12, min_circularity=0.5, sliding_window_radius=2))
    out["hy_detect_alt"] = np.stack(res2)
    # cluster filter on blobby masks
    blob = np.zeros((32, 48), bool)
    brng = np.random.default_rng(5)
    for _ in range(14):
        y, x = int(brng.integers(2, 30)), int(brng.integers(2, 46))
        for _ in range(int(brng.integers(1, 30))):
            blob[min(31, max(0, y + int(brng.integers(-2, 3)))), min(47, max(0, x + int(brng.integers(-3, 4))))] = True
    bframe = brng.integers(0, 256, blob.shape).astype(np.uint8)
    out["cl_mask"], out["cl_frame"] = blob, bframe
    for tag, (mn, mx, mc) in dict(a=(5, 20, 0.1), b=(2, 30, 0.6), c=(10, 20, 0.9)).items():
        cm, cnt, sz, br = ref.filter_damaged_pixel_clusters(bframe, blob, mn, mx, mc)
        out[f"cl_{tag}_mask"] = cm
        out[f"cl_{tag}_stats"] = np.array([cnt, sz, br], float)

    np.savez_compressed(os.path.join(HERE, "reference_run.npz"), **out)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "reference_run.npz")), "bytes")
    print("flows", np.round(flows, 3))
    print("detect", res[0])
    print("clusters", res[1], res[2], res[3])


if __name__ == "__main__":
    main()

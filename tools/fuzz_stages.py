#!/usr/bin/env python3
"""Randomised parity of the two labelling stages through the drop-in module's own functions, on masks the
whole-path sweep never produces (noise of any density, lines, rings, solid and hollow shapes, big clusters):
    filter_damaged_pixel_clusters  (close + CCL + size filter + contour perimeter / circularity + means)
    remove_bright_regions          (two CCLs + overlap rules)
and of compute_background / _find_background / get_damaged_pixel_mask on frames of any content (white noise,
sparse extremes, constants; window radius 1..8), against oracle/stages.py: bit-exact masks and float32
backgrounds / thresholds, equal statistics; and (temporal_case) of the temporal stages and estimates on
arbitrary mask lists with None entries.
    python tools/fuzz_stages.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def random_mask(rng, h, w):
    m = np.zeros((h, w), bool)
    kind = int(rng.integers(0, 5))
    if kind in (0, 4):
        m |= rng.random((h, w)) < float(rng.choice([0.002, 0.01, 0.05, 0.2, 0.45, 0.7]))
    if kind in (1, 4):
        for _ in range(int(rng.integers(1, 25))):            # solid / hollow boxes and ellipses
            cy, cx = int(rng.integers(0, h)), int(rng.integers(0, w))
            ry, rx = int(rng.integers(1, 14)), int(rng.integers(1, 14))
            yy, xx = np.ogrid[:h, :w]
            if rng.random() < 0.5:
                shape = (abs(yy - cy) <= ry) & (abs(xx - cx) <= rx)
                inner = (abs(yy - cy) < ry - 1) & (abs(xx - cx) < rx - 1)
            else:
                shape = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
                inner = ((yy - cy) / max(ry - 1.5, 0.5)) ** 2 + ((xx - cx) / max(rx - 1.5, 0.5)) ** 2 <= 1.0
            m |= (shape & ~inner) if rng.random() < 0.4 else shape
    if kind in (2, 4):
        for _ in range(int(rng.integers(1, 20))):            # lines of every slope
            y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
            dy, dx = int(rng.integers(-12, 13)), int(rng.integers(-12, 13))
            steps = max(abs(dy), abs(dx), 1)
            for t in range(steps + 1):
                y, x = y0 + round(dy * t / steps), x0 + round(dx * t / steps)
                if 0 <= y < h and 0 <= x < w:
                    m[y, x] = True
    if kind == 3:
        m |= rng.random((h, w)) < 0.03                       # noise, grown
        g = m.copy()
        g[1:] |= m[:-1]; g[:, 1:] |= m[:, :-1]
        if rng.random() < 0.5:
            g[:-1] |= m[1:]; g[:, :-1] |= m[:, 1:]
        m = g
    if rng.random() < 0.3:                                   # something on the border
        m[0, :] |= rng.random(w) < 0.3; m[:, -1] |= rng.random(h) < 0.3
    return m


def one_case(ip, seed):
    from oracle import stages as S
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(3, 150)), int(rng.integers(3, 200))
    frame = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mask = random_mask(rng, h, w)
    lo = int(rng.integers(1, 6)); hi = int(rng.choice([5, 20, 60, 400, 5000])); circ = float(rng.choice([0.0, 0.2, 0.5, 0.785, 0.9]))
    cst = int(rng.choice([1, 4, 10, 10, 10, 25]))
    ref = S.filter_damaged_pixel_clusters(frame, mask, lo, hi, circ, cst)
    got = ip.filter_damaged_pixel_clusters(frame, mask, lo, hi, circ, cst)
    assert np.array_equal(got[0], ref[0]), "cluster mask"
    assert got[1] == ref[1], "cluster count %r vs %r" % (got[1], ref[1])
    assert got[2] == ref[2], "mean area %r vs %r" % (got[2], ref[2])
    assert (np.isnan(got[3]) and np.isnan(ref[3])) or got[3] == ref[3], "mean brightness %r vs %r" % (got[3], ref[3])
    # remove_bright_regions: a smooth background with saturated patches, another random mask
    bg = rng.uniform(0, 160, (h, w)).astype(np.float32)
    for _ in range(int(rng.integers(0, 6))):
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        bg[y:y + int(rng.integers(1, 30)), x:x + int(rng.integers(1, 30))] = float(rng.uniform(150, 255))
    dm = random_mask(rng, h, w)
    bt = float(rng.choice([120, 170, 200])); mcs = int(rng.choice([3, 20, 80]))
    r2 = S.remove_bright_regions(bg, bt, dm, mcs)
    g2 = ip.remove_bright_regions(bg, bt, dm, mcs)
    assert np.array_equal(np.asarray(g2, bool), np.asarray(r2, bool)), "remove_bright_regions"
    # background + raw mask on frames of any content (the scenes of the whole-path sweep are smooth):
    # white noise, sparse extremes, constants, +-1 jitter; window radius 1..8, several outlier coefficients
    n = int(rng.integers(2, 20)); radius = int(rng.integers(1, 9))
    hh, ww = int(rng.integers(1, 70)), int(rng.integers(1, 100))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        fr = rng.integers(0, 256, (n, hh, ww), dtype=np.uint8)
    elif kind == 1:
        fr = np.where(rng.random((n, hh, ww)) < 0.1, rng.choice([1, 254, 255], (n, hh, ww)), 0).astype(np.uint8)
    elif kind == 2:
        fr = np.full((n, hh, ww), int(rng.integers(0, 256)), np.uint8)
    else:
        fr = np.clip(int(rng.integers(0, 256)) + rng.integers(-1, 2, (n, hh, ww)), 0, 255).astype(np.uint8)
    i = int(rng.integers(0, n))
    b_ref = S.compute_background(fr, i, radius)
    b_got = ip.compute_background(fr, i, radius)
    assert b_got.dtype == np.float32 and np.array_equal(b_got, b_ref), "compute_background (radius %d, kind %d)" % (radius, kind)
    idx = S.background_window(n, i, radius)
    coeff = float(rng.choice([0.0, 0.5, 1.0, 2.0, -1.0]))
    f_ref = S.compute_background(fr, i, radius, coeff)
    f_got = ip._find_background(fr[idx].astype(np.float32), coeff)
    assert np.array_equal(f_got, f_ref), "_find_background (coeff %g, %d neighbours)" % (coeff, len(idx))
    m_ref, t_ref = S.get_damaged_pixel_mask(fr[i], b_ref)
    m_got, t_got = ip.get_damaged_pixel_mask(fr[i], hh, ww, b_ref)
    assert np.array_equal(m_got, m_ref) and np.array_equal(t_got, t_ref), "get_damaged_pixel_mask"
    return "ok  %3dx%-3d mask %5d px -> %3d clusters (size %d..%d, circ %.2f @%d); bright regions removed %d px" % (
        w, h, int(mask.sum()), ref[1], lo, hi, circ, cst, int(dm.sum()) - int(np.asarray(r2, bool).sum()))


def _same_list(got, ref):
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert (g is None) == (r is None), "None pattern"
        if r is not None:
            assert np.array_equal(np.asarray(g, bool), np.asarray(r, bool)), "mask"


def _same_f(got, ref, what):
    assert np.array_equal(np.asarray(got, float), np.asarray(ref, float), equal_nan=True), what


def temporal_case(ip, seed):
    """The temporal stages on arbitrary mask lists (dense masks, None entries anywhere but frame 0)."""
    from oracle import stages as S
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 30)); h = int(rng.integers(1, 50)); w = int(rng.integers(1, 140))
    frames = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(n)]
    dens = float(rng.choice([0.0, 0.02, 0.3, 0.6, 1.0]))
    base = rng.random((h, w)) < dens                          # pixels that tend to stay lit: runs and static pixels
    masks = [None if (i > 0 and rng.random() < 0.2) else ((rng.random((h, w)) < dens) | (base & (rng.random((h, w)) < 0.8)))
             for i in range(n)]
    T = int(rng.integers(0, 6))
    p_ref = S.compute_persistent_mask(masks, T)
    p_got = ip.compute_persistent_mask(masks, T)
    assert np.array_equal(np.asarray(p_got, bool), p_ref), "compute_persistent_mask T=%d" % T
    f_ref, c_ref = S.filter_consecutive_pixels(masks, p_ref)
    f_got, c_got = ip.filter_consecutive_pixels(masks, p_ref)
    _same_list(f_got, f_ref); _same_f(c_got, c_ref, "filter_consecutive_pixels counts")
    bt = float(rng.choice([100, 170, 250])); st = float(rng.choice([0, 5, 20, 50, 99])); mv = int(rng.choice([0, 3, 10, 40]))
    s_ref = S.compute_static_mask(frames, masks, bt, st, mv)
    s_got = ip.compute_static_mask(frames, masks, bt, st, mv)
    assert np.array_equal(np.asarray(s_got, bool), s_ref), "compute_static_mask"
    a_ref, ac_ref = S.apply_static_suppression(masks, p_ref, s_ref)
    a_got, ac_got = ip.apply_static_suppression(masks, p_ref, s_ref)
    _same_list(a_got, a_ref); _same_f(ac_got, ac_ref, "apply_static_suppression counts")
    thr = int(rng.choice([1, 170, 170, 200, 256]))
    _same_f(ip.estimate_damaged_pixels_in_bright_areas(frames, a_ref, thr), S.estimate_damaged_pixels_in_bright_areas(frames, a_ref, thr), "estimates")
    e_ref = S.find_bright_area_estimates(frames, a_ref, bt)
    _same_f(ip.find_bright_area_estimates(frames, a_ref, bt), e_ref, "find_bright_area_estimates")
    _same_f(ip._get_final_count(a_ref, e_ref), S.get_final_count(a_ref, e_ref), "_get_final_count")
    return "ok"


def main():
    import image_processing as ip
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    t0 = time.time()
    for k in range(cases):
        try:
            msg = one_case(ip, seed0 + k)
            temporal_case(ip, seed0 + k)
        except AssertionError as e:
            print("seed %d: MISMATCH %s" % (seed0 + k, e), flush=True)
            sys.exit(1)
        if k % 20 == 0 or k == cases - 1:
            print("seed %d: %s  [%.0f s]" % (seed0 + k, msg, time.time() - t0), flush=True)
    print("%d cases passed" % cases)


if __name__ == "__main__":
    main()

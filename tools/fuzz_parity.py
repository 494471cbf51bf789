#!/usr/bin/env python3
"""Randomised parity sweep of the whole path against the CPU oracle (a checker, like tests/): random frame
sizes (odd ones included), chunk lengths, parameter sets and scene contents.  Run on the GPU box:
    python tools/fuzz_parity.py [cases] [seed] [max_side] [wide | exact8]
Prints one line per case and exits non-zero at the first mismatch (the case's seed reproduces it)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOW_RTOL = 1e-5


def eq_nan(a, b):
    return np.array_equal(np.asarray(a, float), np.asarray(b, float), equal_nan=True)


def one_case(eng, seed, max_side=260, wide=False):
    from image_processing_amd import synth
    from oracle import pipeline
    rng = np.random.default_rng(seed)
    h = int(rng.integers(34, max(35, int(max_side * 0.77)))); w = int(rng.integers(34, max_side)); n = int(rng.integers(2, 40))
    params = dict(consecutive_threshold=int(rng.integers(1, 5)), brightness_threshold=int(rng.integers(120, 220)),
                  flow_threshold=float(rng.choice([0.05, 0.5, 2.0, 4.0])), static_threshold=float(rng.choice([5, 20, 50])),
                  min_cluster_size=int(rng.integers(1, 4)), max_cluster_size=int(rng.integers(6, 40)),
                  min_circularity=float(rng.choice([0.0, 0.3, 0.6])), sliding_window_radius=int(rng.integers(1, 6)))
    if wide:                                                 # the corners of the parameter space
        n = int(rng.integers(2, 90))
        params.update(consecutive_threshold=int(rng.integers(0, 8)), sliding_window_radius=int(rng.integers(1, 9)),
                      max_cluster_size=int(rng.choice([1, 5, 40, 300])), min_cluster_size=int(rng.integers(0, 7)),
                      min_circularity=float(rng.choice([0.0, 0.785, 1.0])), brightness_threshold=int(rng.choice([0, 100, 170, 254, 255])),
                      static_threshold=float(rng.choice([0, 1, 50, 100])))
    motion = tuple(int(x) for x in rng.choice(np.arange(1, max(2, n)), size=min(int(rng.integers(0, 4)), max(1, n - 1)), replace=False)) if n > 2 else ()
    frames = synth.chunk_numpy(seed, n, h, w, hot=float(rng.uniform(2, 30)), stuck=int(rng.integers(0, 6)),
                               bright_block=bool(rng.integers(0, 2)), motion_at=motion, shift=float(rng.uniform(0.5, 9)),
                               blobs=int(rng.integers(0, 6)))
    (total, cc, sz, br), mid = pipeline.detect_damaged_pixels(frames, params=params, return_intermediates=True)
    eng.set_params(params)
    try:
        r = eng.detect_chunk(frames, want_masks=True)
    finally:
        eng.set_params(None)
    thr = params["flow_threshold"]
    near = np.abs(mid["flows"] - thr) <= 1e-4 * thr
    if near.any():
        return "skipped (a flow metric within 1e-4 of the threshold)"
    if not np.allclose(r["flows"], mid["flows"], rtol=FLOW_RTOL, atol=1e-12):
        got, ref = np.asarray(r["flows"], float), np.asarray(mid["flows"], float)
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
        i = int(np.argmax(rel))
        raise AssertionError("flows: %dx%d n=%d frame %d got %.9g ref %.9g rel %.3g (motion_at=%s)" % (w, h, n, i, got[i], ref[i], rel[i], motion))
    assert np.array_equal(r["keep"].astype(bool), mid["flows"] <= thr), "keep"
    for i in range(n):
        if mid["final"][i] is None:
            assert r["final_masks"][i] is None, "final mask %d should be None" % i
        else:
            assert np.array_equal(r["final_masks"][i], mid["final"][i]), "final mask %d" % i
    assert eq_nan(r["total_counts"], total), "total counts"
    assert eq_nan(r["cluster_counts"], cc) and eq_nan(r["avg_sizes"], sz) and eq_nan(r["avg_brightnesses"], br), "cluster stats"
    return "ok  %3dx%-3d n=%-2d kept=%-2d counts=%d clusters=%d r=%d T=%d" % (
        w, h, n, int(np.sum(r["keep"])), int(np.nansum(total)), int(np.nansum(cc)), params["sliding_window_radius"],
        params["consecutive_threshold"])


def flow_case(eng, seed, exact8=False):
    """One Farneback pair of arbitrary content (white noise, constants, steps, shifted textures; sizes from 2x2,
    so 1 to 4 pyramid levels and every resize path) against the oracle's field, value by value.
    exact8: sizes that are multiples of 8 from 256 up (levels 2 and 3 are then made by pyr_exact_kernel)."""
    from oracle import cref
    rng = np.random.default_rng(seed)
    h = int(rng.integers(2, 300)); w = int(rng.integers(2, 400))
    if exact8:
        h = 8 * int(rng.integers(32, 72)); w = 8 * int(rng.integers(32, 90))
    kind = int(rng.integers(0, 5))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    if kind == 0:
        a = rng.integers(0, 256, (h, w)); b = rng.integers(0, 256, (h, w))
    elif kind == 1:
        a = np.full((h, w), int(rng.integers(0, 256))); b = np.full((h, w), int(rng.integers(0, 256)))
    elif kind == 2:
        a = (xx > w / 2) * 255; b = (xx > w / 2 + float(rng.uniform(-9, 9))) * 255
    else:
        dx, dy = float(rng.uniform(-12, 12)), float(rng.uniform(-12, 12))
        def tex(ox, oy):
            v = 110.0 + np.zeros((h, w))
            for kx, ky, ph in ((0.61, 0.23, 0.3), (0.17, 0.55, 1.1), (-0.41, 0.37, 2.0), (0.29, -0.47, 0.7)):
                v += 14 * np.sin((xx - ox) * kx + (yy - oy) * ky + ph)
            return v + rng.normal(0, float(rng.choice([0, 2, 10])), (h, w))
        a = tex(0, 0); b = tex(dx, dy)
    a = np.clip(np.rint(a), 0, 255).astype(np.float32); b = np.clip(np.rint(b), 0, 255).astype(np.float32)
    ref = cref.farneback(a, b)
    got = eng.farneback_pair(eng.to_device(a), eng.to_device(b)).cpu().numpy()
    assert np.isfinite(got).all() == np.isfinite(ref).all(), "finiteness"
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    scale = max(1.0, float(np.abs(ref).max()))
    # (flat regions: both fields are zero up to residues of ~1e-30 that depend on the order of the f64 sums)
    diff = err > 1e-12 * scale
    if diff.any():
        assert err.max() <= 1e-6 * scale and diff.sum() <= max(1, 1e-5 * diff.size), \
            "flow field %dx%d kind %d: %d of %d values differ, max %.3g (scale %.3g)" % (w, h, kind, diff.sum(), diff.size, err.max(), scale)
    return "ok  %3dx%-3d kind %d: %d values differ" % (w, h, kind, int(diff.sum()))


def main():
    from image_processing_amd.engine import Engine
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    max_side = int(sys.argv[3]) if len(sys.argv) > 3 else 260
    wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
    eng = Engine(0)
    t0 = time.time()
    if len(sys.argv) > 4 and sys.argv[4] == "exact8":         # flow fields only, at sizes that take pyr_exact_kernel
        for k in range(cases):
            try:
                print("seed %d: flow %s  [%.0f s]" % (seed0 + k, flow_case(eng, seed0 + k, True), time.time() - t0), flush=True)
            except AssertionError as e:
                print("seed %d: MISMATCH %s" % (seed0 + k, e), flush=True)
                sys.exit(1)
        eng.close()
        print("%d cases passed" % cases)
        return
    for k in range(cases):
        seed = seed0 + k
        try:
            msg = one_case(eng, seed, max_side, wide)
            fmsg = flow_case(eng, seed)
            if not fmsg.endswith(": 0 values differ"):
                print("seed %d: flow %s" % (seed, fmsg), flush=True)
        except AssertionError as e:
            print("seed %d: MISMATCH %s" % (seed, e), flush=True)
            sys.exit(1)
        print("seed %d: %s  [%.0f s]" % (seed, msg, time.time() - t0), flush=True)
    eng.close()
    print("%d cases passed" % cases)


if __name__ == "__main__":
    main()

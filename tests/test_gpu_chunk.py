"""The motion test (Farneback flow metric) and the whole-chunk entry point vs the CPU oracle and
the golden outputs of the reference's own orchestration.  Needs an MI355X: ``-m gpu``.

Tolerances (north_star: bit-exact masks / discard lists / counts, 1e-5 relative float metrics):
* flow field: per-pixel expressions, the vertical 15-row running sums (including their f32-rounded updates) and
  the solve are evaluated exactly as the oracle does; only the horizontal 15-sums are formed in a different
  order, in f64, so a field value can differ from the oracle's by one f32 ulp at most, and rarely: the test
  allows 1e-5 of the values to differ, by no more than 1e-6 of the flow scale (measured: none differ);
* the mean magnitude is an f64 tree against numpy's f32 pairwise sum: 1e-5 relative;
* everything downstream of the discard list is compared bit-exactly.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FLOW_RTOL = 1e-5


@pytest.fixture(scope="module")
def eng():
    from image_processing_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a, float), np.asarray(b, float), equal_nan=True)


def _moving_pair(seed, h, w, shift):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)

    def scene(off):
        x = xx - off
        return 90 + 50 * np.sin(x / 9.0) * np.cos(yy / 7.0) + 30 * np.sin((x + 2 * yy) / 13.0)
    a = np.clip(np.rint(scene(0.0) + rng.normal(0, 1.5, (h, w))), 0, 255).astype(np.uint8)
    b = np.clip(np.rint(scene(shift) + rng.normal(0, 1.5, (h, w))), 0, 255).astype(np.uint8)
    return a, b


# (h, w): single level; 3 exact halvings; odd sizes (general linear resize); non-multiple-of-64 widths
@pytest.mark.parametrize("shape,shift", [((40, 56), 1.0), ((264, 360), 3.0), ((131, 203), 0.7), ((270, 480), 6.0),
                                         ((67, 300), 2.0), ((540, 960), 4.0)])
def test_farneback_pair_field(eng, shape, shift):
    from oracle import cref
    h, w = shape
    a, b = _moving_pair(h + w, h, w, shift)
    ref = cref.farneback(a.astype(np.float32), b.astype(np.float32))
    got = eng.farneback_pair(eng.to_device(a.astype(np.float32)), eng.to_device(b.astype(np.float32))).cpu().numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    scale = max(1.0, float(np.abs(ref).max()))
    assert err.max() <= 1e-6 * scale, "max |dflow| = %g (flow scale %g)" % (err.max(), scale)
    ndiff = int((err > 1e-12 * scale).sum())          # (residues of ~1e-30 in flat regions depend on the order of the f64 sums)
    assert ndiff <= 1e-5 * err.size, "%d of %d flow values differ" % (ndiff, err.size)
    mref = float(np.mean(cref.flow_magnitude(ref)))
    mgot = float(np.mean(cref.flow_magnitude(got)))
    assert abs(mgot - mref) <= FLOW_RTOL * max(mref, 1e-30)


# Sizes with H and W multiples of 4: level 0 of the pyramid is then evaluated inside polyexp_kernel<true> (no level-0
# image in HBM).  One tile in each direction (every tile touches a border), tiles whose window leaves the image on
# one side only, interior tiles with 16-byte loads, a width that ends exactly at a tile edge (reflect-101 of column W),
# heights below one tile and below the blur's reflection distance.
@pytest.mark.parametrize("shape,shift", [((4, 4), 0.5), ((8, 36), 1.0), ((16, 64), 1.0), ((36, 72), 2.0), ((64, 128), 1.5),
                                         ((20, 200), 2.0), ((100, 136), 3.0), ((52, 264), 0.7), ((132, 392), 5.0)])
def test_farneback_pair_field_fused_level0(eng, shape, shift):
    from oracle import cref
    h, w = shape
    for seed, noise in ((h * w, True), (h + w, False)):
        if noise:                                      # white noise: every bit of the blur and the expansion matters
            rng = np.random.default_rng(seed)
            a = rng.integers(0, 256, (h, w)).astype(np.uint8); b = rng.integers(0, 256, (h, w)).astype(np.uint8)
        else:
            a, b = _moving_pair(seed, h, w, shift)
        ref = cref.farneback(a.astype(np.float32), b.astype(np.float32))
        got = eng.farneback_pair(eng.to_device(a.astype(np.float32)), eng.to_device(b.astype(np.float32))).cpu().numpy()
        assert got.shape == ref.shape and np.isfinite(got).all()
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        scale = max(1.0, float(np.abs(ref).max()))
        assert err.max() <= 1e-12 * scale, "%dx%d noise=%s: max |dflow| = %g (flow scale %g)" % (w, h, noise, err.max(), scale)


# Sizes with H and W multiples of 8 and at least 256: levels 2 and 3 are then exact quarter / eighth scales and take
# pyr_exact_kernel (aligned 16-byte window reads, no tables).  White noise, so that every bit of the 9- and 19-tap blurs
# counts; widths and heights that end inside a tile, a single tile column at level 3, interior tiles with dword loads.
@pytest.mark.parametrize("shape", [(256, 256), (256, 320), (264, 520), (328, 264), (512, 288)])
def test_farneback_pair_field_exact_pyramid(eng, shape):
    from oracle import cref
    h, w = shape
    rng = np.random.default_rng(h * 1000 + w)
    for noise in (True, False):
        if noise:
            a = rng.integers(0, 256, (h, w)).astype(np.uint8); b = rng.integers(0, 256, (h, w)).astype(np.uint8)
        else:
            a, b = _moving_pair(h + w, h, w, 2.5)
        ref = cref.farneback(a.astype(np.float32), b.astype(np.float32))
        got = eng.farneback_pair(eng.to_device(a.astype(np.float32)), eng.to_device(b.astype(np.float32))).cpu().numpy()
        assert got.shape == ref.shape and np.isfinite(got).all()
        err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        scale = max(1.0, float(np.abs(ref).max()))
        assert err.max() <= 1e-12 * scale, "%dx%d noise=%s: max |dflow| = %g (flow scale %g)" % (w, h, noise, err.max(), scale)


def test_farneback_rejects_non_u8_images(eng):
    from image_processing_amd._ipx import IpxError
    a = np.full((40, 40), 10.5, np.float32)
    with pytest.raises(IpxError):
        eng.farneback_pair(eng.to_device(a), eng.to_device(a))


def test_flow_rejects_frames_beyond_32bit_addressing(eng):
    import torch
    from image_processing_amd._ipx import IpxError
    big = torch.zeros((2, 8200, 8200), dtype=torch.uint8, device="cuda:0")      # > 2^26 pixels
    with pytest.raises(IpxError, match="2\\^26"):
        eng.flow_metric(big)


def test_flow_metric_known_answers(eng):           # t.py:706-727
    z = np.zeros((3, 8, 8), np.uint8)
    flows, keep = eng.flow_metric(eng.to_device(z))
    assert np.all(flows.cpu().numpy() == 0.0) and np.all(keep.cpu().numpy() == 1)
    f0 = np.zeros((10, 10), np.uint8); f1 = f0.copy(); f0[3, 3] = 255; f1[7, 7] = 255
    flows, _ = eng.flow_metric(eng.to_device(np.stack([f0, f1])))
    fl = flows.cpu().numpy()
    assert fl.shape == (2,) and fl[0] == 0.0 and fl[1] > 0.0


@pytest.mark.parametrize("shape", [(96, 128), (131, 203)])
def test_flow_metric_and_keep(eng, shape):
    from image_processing_amd import synth
    from oracle import stages as S
    h, w = shape
    frames = synth.chunk_numpy(21, 12, h, w, motion_at=(4, 5, 9), shift=5.0)
    ref = S.compute_optical_flow_metric(frames)
    flows, keep = eng.flow_metric(eng.to_device(frames))
    got = flows.cpu().numpy().astype(np.float64)
    assert got[0] == 0.0
    assert np.allclose(got, ref, rtol=FLOW_RTOL, atol=1e-12)
    assert np.array_equal(keep.cpu().numpy().astype(bool), ref <= 2.0)
    assert not (ref <= 2.0).all() and (ref <= 2.0).sum() >= 6      # the case exercises both branches
    # left-halo form: flows of frames 5.. given frame 4 as prev_frame
    fd = eng.to_device(frames)
    f2, k2 = eng.flow_metric(fd[5:], prev_frame=fd[4])
    assert np.array_equal(f2.cpu().numpy(), flows.cpu().numpy()[5:])
    assert np.array_equal(k2.cpu().numpy(), keep.cpu().numpy()[5:])


def test_chunk_vs_reference_golden(eng, golden):
    """ip.detect_damaged_pixels itself (run over the oracle's cv2 restatements) on the 16-frame
    40x56 chunk with three motion frames: counts, NaNs and cluster metrics, two parameter sets."""
    ch = golden["hy_frames"]
    r = eng.detect_chunk(ch)
    assert np.allclose(r["flows"], golden["hy_flows"], rtol=FLOW_RTOL, atol=1e-12)
    got = np.stack([r["total_counts"], r["cluster_counts"], r["avg_sizes"], r["avg_brightnesses"]])
    assert _eq_nan(got, golden["hy_detect"])
    alt = dict(consecutive_threshold=3, brightness_threshold=150, flow_threshold=0.05, static_threshold=20,
               min_cluster_size=2, max_cluster_size=12, min_circularity=0.5, sliding_window_radius=2)
    eng.set_params(alt)
    r = eng.detect_chunk(ch)
    got = np.stack([r["total_counts"], r["cluster_counts"], r["avg_sizes"], r["avg_brightnesses"]])
    eng.set_params(None)
    assert _eq_nan(got, golden["hy_detect_alt"])


@pytest.mark.parametrize("shape,n,kw", [
    ((96, 128), 24, dict(motion_at=(7, 8, 15), shift=5.0, hot=12.0, stuck=4)),
    ((131, 203), 16, dict(motion_at=(3,), shift=7.0, hot=20.0, stuck=5, blobs=6)),
    ((72, 70), 13, dict(hot=9.0)),
])
def test_chunk_vs_oracle(eng, shape, n, kw):
    from image_processing_amd import synth
    from oracle import pipeline
    h, w = shape
    frames = synth.chunk_numpy(100 + h, n, h, w, **kw)
    (total, cc, sz, br), mid = pipeline.detect_damaged_pixels(frames, return_intermediates=True)
    r = eng.detect_chunk(frames, want_masks=True)
    assert np.allclose(r["flows"], mid["flows"], rtol=FLOW_RTOL, atol=1e-12)
    assert np.array_equal(r["keep"].astype(bool), mid["flows"] <= 2.0)
    for i in range(n):
        if mid["final"][i] is None:
            assert r["final_masks"][i] is None
        else:
            assert np.array_equal(r["final_masks"][i], mid["final"][i]), "final mask %d" % i
    assert _eq_nan(r["total_counts"], total)
    assert _eq_nan(r["cluster_counts"], cc) and _eq_nan(r["avg_sizes"], sz) and _eq_nan(r["avg_brightnesses"], br)
    assert np.nansum(total) > 0


def test_chunk_is_deterministic_and_matches_stagewise(eng):
    """Run-to-run bitwise equality, and the fused chunk call equals the stage-by-stage calls."""
    from image_processing_amd import synth
    frames = synth.chunk_numpy(77, 20, 120, 200, motion_at=(6,), hot=15.0, stuck=4)
    fd = eng.to_device(frames)
    a = eng.detect_chunk_device(fd, want_final_bits=True)
    b = eng.detect_chunk_device(fd, want_final_bits=True)
    for k in ("flows", "keep", "counts", "cluster", "final_bits"):
        assert bool((a[k] == b[k]).all()), k
    n, h, w = frames.shape
    flows, keep = eng.flow_metric(fd)
    raw, bright, dense = eng.detect_raw(fd, keep=keep)
    raw = eng.remove_bright_regions(bright, raw, h, w)
    pers = eng.persistent_mask(raw, keep, h, w)
    st = eng.static_from_maps(eng.static_maps(fd, raw, pers, keep))
    fin, c3 = eng.finalize(fd, raw, pers, st, keep)
    cl, _ = eng.cluster_stats(fd, fin, keep)
    assert bool((fin == a["final_bits"]).all()) and bool((cl == a["cluster"]).all())
    assert bool((a["counts"][:, :3] == c3).all()) and bool((a["counts"][:, 3:] == dense).all())


def test_async_chunks_overlap_and_match(eng):
    """Two chunks in flight at once (different shapes, so the scratch arenas regrow between them): each
    handle returns exactly what the blocking call returns."""
    from image_processing_amd import synth
    fa = synth.chunk_numpy(5, 18, 96, 128, motion_at=(4,), hot=14.0, stuck=3)
    fb = synth.chunk_numpy(6, 11, 131, 203, hot=10.0, blobs=4)
    pa = eng.detect_chunk_async(fa)
    pb = eng.detect_chunk_async(fb)
    ra, rb = pa.result(), pb.result()
    for got, frames in ((ra, fa), (rb, fb)):
        ref = eng.detect_chunk(frames)
        for k in ("total_counts", "cluster_counts", "avg_sizes", "avg_brightnesses", "flows"):
            assert _eq_nan(got[k], ref[k]), k
        assert np.array_equal(got["keep"], ref["keep"]) and np.array_equal(got["counts"], ref["counts"])


def test_randomised_parity_sweep(eng):
    """tools/fuzz_parity.py: random sizes (odd ones included), chunk lengths, parameter sets and scenes, the
    whole path against the oracle, bit-exact downstream of the flow metric."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_parity
    done = 0
    for seed in range(7000, 7150):
        msg = fuzz_parity.one_case(eng, seed)
        done += msg.startswith("ok")
    assert done >= 140
    for seed in range(7000, 7080):                          # single pairs of any content, field against field
        assert fuzz_parity.flow_case(eng, seed).startswith("ok")


def test_flow_handoff_timeout_is_not_silent():
    """flow_iter_kernel's consumer waves meet on an LDS counter with a bounded wait.  With the bound forced to
    zero polls (ipx_debug_fi_spin_limit) the wait runs out: the call must then report NaN metrics with keep = 0,
    ipx_synchronize must fail, detect_chunk must raise, and after clearing the condition the same context must
    give the right answer again."""
    import torch
    from image_processing_amd import _ipx, synth
    from image_processing_amd.engine import Engine
    frames = synth.chunk_numpy(3, 9, 120, 300, hot=5.0, stuck=1, motion_at=(4,), shift=5.0)
    ref = Engine(0)
    good, good_keep = ref.flow_metric(ref.to_device(frames))
    good, good_keep = good.cpu().numpy(), good_keep.cpu().numpy()
    ref.close()
    eng = Engine(0)
    try:
        fr = eng.to_device(frames)
        _ipx.check(_ipx.lib.ipx_debug_fi_spin_limit(eng._ctx, 0))
        flows, keep = eng.flow_metric(fr)
        torch.cuda.synchronize()
        flows, keep = flows.cpu().numpy(), keep.cpu().numpy()
        assert flows[0] == 0 and keep[0] == 1                      # frame 0 has no pair
        assert np.isnan(flows[1:]).all() and not keep[1:].any()
        with pytest.raises(_ipx.IpxError) as ei:
            eng.synchronize()
        assert ei.value.status == -2 and "timed out" in str(ei.value)
        with pytest.raises(RuntimeError):
            eng.detect_chunk(fr)
        assert eng.flow_status(clear=True) is True
        _ipx.check(_ipx.lib.ipx_debug_fi_spin_limit(eng._ctx, 1 << 27))
        assert eng.flow_status() is False
        flows, keep = eng.flow_metric(fr)
        eng.synchronize()
        assert np.array_equal(flows.cpu().numpy(), good) and np.array_equal(keep.cpu().numpy(), good_keep)
    finally:
        eng.close()


def test_upload_frames_side_stream_then_detect():
    """ipx_host_register + ipx_upload_frames: host frames reach HBM on the context's side stream and the calls that
    follow on the context see them; the result equals the one from a plain torch upload."""
    import torch
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine
    frames = synth.chunk_numpy(17, 12, 72, 100, hot=9.0, stuck=2, motion_at=(5,), shift=6.0)
    eng = Engine(0)
    try:
        ref = eng.detect_chunk(frames)
        for register in (True, False):
            up = eng.upload_frames(frames.copy(), register=register)
            got = eng.detect_chunk(up.tensor)
            up.release()
            for k in ("total_counts", "cluster_counts", "avg_sizes", "avg_brightnesses", "flows", "keep"):
                assert np.array_equal(got[k], ref[k], equal_nan=True), (register, k)
            assert bool((up.tensor.cpu() == torch.from_numpy(frames)).all())
    finally:
        eng.close()


def test_flow_above_25_megapixels(eng):
    """a SHORT call on frames above 25.2 M pixels.  The dead positions of flow_iter_kernel's loads used to
    resolve, for the 4-byte coefficient plane, to an offset inside such a frame's resource and from there (scalar offset
    added) to up to 0.9 GB past the end of a small batch's allocation.  One pair (the least scratch, the case with the
    least slack behind the arena) of 6144 x 4320 frames: the field and the metric must equal the oracle's as at any other
    size, and the stage's error word stays clear."""
    from oracle import cref
    h, w = 4320, 6144                                   # 26.5 M pixels
    a, b = _moving_pair(5, h, w, 3.0)
    ref = cref.farneback(a.astype(np.float32), b.astype(np.float32))
    got = eng.farneback_pair(eng.to_device(a.astype(np.float32)), eng.to_device(b.astype(np.float32))).cpu().numpy()
    eng.synchronize()
    assert not eng.flow_status()
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(got).all() and err.max() <= 1e-6 * scale, "max |dflow| = %g (flow scale %g)" % (err.max(), scale)
    assert int((err > 1e-12 * scale).sum()) <= 1e-5 * err.size
    mref = float(np.mean(cref.flow_magnitude(ref)))
    del got, err
    flows, keep = eng.flow_metric(eng.to_device(np.stack([a, b])))
    eng.synchronize()
    assert not eng.flow_status()
    f = flows.cpu().numpy()
    assert f[0] == 0.0 and abs(f[1] - mref) <= FLOW_RTOL * mref and keep.cpu().numpy().tolist() == [1, 0]
    import torch
    torch.cuda.empty_cache()


def test_flow_error_word_survives_a_bigger_call(eng):
    """the stage's error word is sticky until ipx_flow_status(..., clear) -- also when a later, larger call
    makes the arena that holds it grow."""
    import ctypes as C
    from image_processing_amd import _ipx, synth
    from image_processing_amd.engine import Engine
    e2 = Engine(0)
    try:
        small = e2.to_device(synth.chunk_numpy(3, 3, 40, 56, hot=4.0))
        _ipx.check(_ipx.lib.ipx_debug_fi_spin_limit(e2._ctx, 0))       # every hand-off gives up at once
        flows, keep = e2.flow_metric(small)
        assert np.isnan(flows.cpu().numpy()[1:]).all() and keep.cpu().numpy()[1:].sum() == 0
        _ipx.check(_ipx.lib.ipx_debug_fi_spin_limit(e2._ctx, 1 << 27))
        big = e2.to_device(synth.chunk_numpy(4, 40, 300, 420, hot=4.0))   # needs a larger reduction arena
        e2.flow_metric(big)
        assert e2.flow_status(clear=True)                                  # still raised after the regrow
        flows, _ = e2.flow_metric(big)
        assert not e2.flow_status() and np.isfinite(flows.cpu().numpy()).all()
    finally:
        e2.close()


@pytest.mark.gpu
def test_reserve_flow_changes_nothing():
    """ipx_reserve_flow: plan and arenas ahead of the first call; a smaller and a larger reservation, then the metric."""
    import torch
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine
    fr = synth.chunk_torch(11, 9, 70, 94, torch.device("cuda", 0))
    ref, _ = Engine(0).flow_metric(fr)
    eng = Engine(0)
    eng.reserve_flow(4, 70, 94)
    eng.reserve_flow(40, 70, 94)
    got, _ = eng.flow_metric(fr)
    assert torch.equal(ref, got)
    eng.reserve_flow(3, 40, 52)                       # another geometry in between: the plan is rebuilt on demand
    got2, _ = eng.flow_metric(fr)
    assert torch.equal(ref, got2)
    with pytest.raises(Exception):
        eng.reserve_flow(0, 70, 94)


@pytest.mark.gpu
def test_builtin_profiler_classes_and_flow_levels():
    """ipx_profile_*: per kernel class launches / milliseconds / credited bytes of a chunk, and the flow_iter class
    split by pyramid level (three iterations per level and batch)."""
    import torch
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine
    eng = Engine(0)
    fr = synth.chunk_torch(12, 9, 136, 200, torch.device("cuda", 0))
    eng.detect_chunk(fr)
    eng.profile_enable(True)
    eng.profile_reset()
    eng.detect_chunk(fr)
    prof, levels = eng.profile_read(), eng.profile_read_flow_levels()
    eng.profile_enable(False)
    assert {"flow_iter", "polyexp", "detect"} <= set(prof)
    assert all(v["launches"] > 0 and v["ms"] > 0 and v["bytes"] >= 0 for v in prof.values())
    assert sorted(levels) == [0, 1, 2] and all(d["launches"] == 3 for d in levels.values())      # 136 x 200: two coarser levels
    assert sum(d["launches"] for d in levels.values()) == prof["flow_iter"]["launches"]
    assert abs(sum(d["ms"] for d in levels.values()) - prof["flow_iter"]["ms"]) < 1e-6 * max(1.0, prof["flow_iter"]["ms"])
    assert levels[0]["bytes"] > levels[1]["bytes"] > levels[2]["bytes"] > 0
    eng.profile_reset()
    assert eng.profile_read() == {} and eng.profile_read_flow_levels() == {}


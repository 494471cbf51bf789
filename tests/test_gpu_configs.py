"""BASELINE.json's configurations as parity cases (``-m gpu``).

* configs[0]  640x480, 300 frames, injected hot pixels, static camera: the reference's own CPU-runnable
  case, compared with the oracle over the whole chunk (bit-exact integers, 1e-5 relative flow metric);
* configs[1]  1920x1080: a slice the oracle finishes in seconds is compared exactly; the full 1000-frame
  chunk is covered by size-independent properties (run-to-run bitwise determinism, sharded == unsharded,
  a chunk equals the concatenation of its frame-independent parts, counts invariant under a frame shuffle
  where the path is frame-independent);
* configs[4]  3840x2160: a slice vs the oracle (frame dimensions, not throughput);
* configs[3]  camera-motion-heavy sequence: the discard list and everything downstream of it.
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


def _vs_oracle(eng, frames):
    from oracle import pipeline
    (total, cc, sz, br), mid = pipeline.detect_damaged_pixels(frames, return_intermediates=True)
    r = eng.detect_chunk(frames, want_masks=True)
    assert np.allclose(r["flows"], mid["flows"], rtol=FLOW_RTOL, atol=1e-12)
    thr = 2.0
    safe = np.abs(mid["flows"] - thr) > 1e-4 * thr               # frames whose metric is not at the threshold
    assert np.array_equal(r["keep"].astype(bool)[safe], (mid["flows"] <= thr)[safe])
    assert np.array_equal(r["keep"].astype(bool), mid["flows"] <= thr), "a metric within 1e-4 of the threshold flipped"
    for i in range(len(frames)):
        if mid["final"][i] is None:
            assert r["final_masks"][i] is None
        else:
            assert np.array_equal(r["final_masks"][i], mid["final"][i]), "final mask %d" % i
    assert _eq_nan(r["total_counts"], total) and _eq_nan(r["cluster_counts"], cc)
    assert _eq_nan(r["avg_sizes"], sz) and _eq_nan(r["avg_brightnesses"], br)
    return r, mid


def test_config0_640x480_300_frames_static_camera(eng):
    from image_processing_amd import synth
    frames = synth.chunk_numpy(300, 300, 480, 640, hot=40.0, stuck=6, blobs=3)
    r, mid = _vs_oracle(eng, frames)
    assert r["keep"].all() and np.nansum(r["total_counts"]) > 0 and np.nansum(r["cluster_counts"]) > 0


def test_config1_1080p_slice_vs_oracle(eng):
    from image_processing_amd import synth
    # 70 frames: two batches of frame pairs in the flow stage, ~2.3 M bit-plane words per labelling pass
    frames = synth.chunk_numpy(1080, 70, 1080, 1920, hot=120.0, stuck=8, blobs=4, motion_at=(5, 40), shift=9.0)
    r, mid = _vs_oracle(eng, frames)
    assert not r["keep"][5] and not r["keep"][40] and r["keep"][[0, 1, 2, 3, 4]].all()


def test_config4_4k_slice_vs_oracle(eng):
    from image_processing_amd import synth
    # few clusters: the oracle's cluster filter costs one full-frame pass per label, as the reference's does
    frames = synth.chunk_numpy(2160, 5, 2160, 3840, hot=12.0, stuck=3, blobs=2, motion_at=(3,), shift=12.0)
    r, mid = _vs_oracle(eng, frames)
    assert not r["keep"][3] and np.nansum(r["total_counts"]) > 0


def test_config3_camera_motion_heavy(eng):
    from image_processing_amd import synth
    # scene jumps of several sizes: sub-threshold drift, clear motion, and bursts of consecutive motion frames
    frames = synth.chunk_numpy(33, 48, 270, 480, hot=25.0, stuck=5, blobs=8,
                               motion_at=(4, 9, 10, 11, 20, 30, 31, 40), shift=5.0)
    r, mid = _vs_oracle(eng, frames)
    dropped = np.flatnonzero(~r["keep"].astype(bool))
    assert set(dropped) == {4, 9, 10, 11, 20, 30, 31, 40}
    from image_processing_amd import hostmath
    mg = hostmath.flow_margin(r["flows"], 2.0)
    print("config3 (jumps) flow margin:", mg)
    # the margin is far above both the float tolerance of the metric (1e-5 relative) and the oracle/GPU difference
    assert mg["min_margin"] > 1e-2 and mg["frames_within_rel"] == 0
    assert np.array_equal(r["keep"].astype(bool), mid["flows"] <= 2.0)


def test_config3_1080p_known_translations(eng):
    """The bench's --motion workload at full size (size-independent property: ground truth).  The scene moves
    by 0, 0.5, 4 and 8 px/frame in 50-frame runs: the flow metric must recover those speeds (2 %), the runs
    faster than flow_threshold = 2 are discarded, and discarded frames report NaN counts / zero clusters."""
    import torch
    from image_processing_amd import synth
    fr = synth.chunk_torch(7, 200, 1080, 1920, torch.device("cuda", 0), speeds=[0, 0.5, 4, 8])
    r = eng.detect_chunk(fr)
    flows, keep = np.asarray(r["flows"]), np.asarray(r["keep"]).astype(bool)
    for seg, v in enumerate([0.0, 0.5, 4.0, 8.0]):
        f = flows[50 * seg + 1:50 * seg + 50] if seg == 0 else flows[50 * seg:50 * seg + 50]
        if v == 0.0:
            assert f.max() < 0.1                      # sensor noise only
        else:
            assert np.all(np.abs(f - v) <= 0.02 * v), (v, f.min(), f.max())
    assert keep[:100].all() and not keep[100:].any()
    # how close the motion test comes to its threshold (ip.py:559): the discard list rests on float metrics that
    # agree with the reference's to ~1e-7 relative, so a margin of this size cannot flip a frame
    from image_processing_amd import hostmath
    mg = hostmath.flow_margin(flows, 2.0)
    print("config3 flow margin:", mg)
    assert mg["min_margin"] > 1.4 and mg["frames_within_rel"] == 0
    assert np.isnan(r["total_counts"][100:]).all() and np.isfinite(r["total_counts"][:100]).all()
    assert np.all(np.asarray(r["cluster_counts"])[100:] == 0)


def test_config1_full_chunk_properties(eng):
    """1000 frames of 1080p (the bench workload): bitwise determinism, sharded == whole, and the flow metric
    of a sub-range equals the same frames' metric inside the whole chunk (pairs are independent)."""
    import torch
    from image_processing_amd import sharding, synth
    dev = torch.device("cuda", 0)
    fr = synth.chunk_torch(2, 1000, 1080, 1920, dev)
    a = eng.detect_chunk_device(fr)
    b = eng.detect_chunk_device(fr)
    for k in ("flows", "keep", "counts", "cluster"):
        assert bool((a[k] == b[k]).all()), k
    assert int(a["keep"].sum()) > 0 and int(a["counts"][:, 0].sum()) > 0
    # the flow metric of frames 400..463 computed alone (different batch boundaries, different launch shapes)
    sub = fr[399:464]
    fl, _ = eng.flow_metric(sub)
    assert bool((fl[1:] == a["flows"][400:464]).all())
    # frame-range sharding (two shards as threads on this GPU) reproduces the whole-chunk result
    import threading
    from image_processing_amd.engine import Engine
    from test_gpu_sharding import ThreadComm
    n, world = 1000, 2
    comm = ThreadComm(world)
    res, errors = [None] * world, []

    def work(rank):
        try:
            e = Engine(0)
            lo_, hi_ = sharding.shard_range(n, rank, world)
            lo, hi = sharding.halo_range(n, lo_, hi_, int(e.params.sliding_window_radius))
            out = sharding.detect_chunk_sharded(e, fr[lo:hi].contiguous(), lo_, hi_, n, comm.view(rank))
            torch.cuda.synchronize()
            res[rank] = {k: out[k].clone() for k in ("flows", "keep", "counts", "cluster")}
            e.close()
        except Exception as ex:                     # noqa: BLE001
            errors.append(ex)
            comm.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not errors, errors
    for r in range(world):
        for k in ("flows", "keep", "counts", "cluster"):
            assert bool((res[r][k].to(a[k].dtype) == a[k]).all()), (r, k)


def test_config4_4k_full_chunk_properties(eng):
    """configs[4] at full size on one GPU: a 1000-frame chunk of 3840x2160 (the unit each of the 8 GPUs of the
    multi-video batch processes).  Size-independent properties: bitwise determinism, and frame-range sharding
    (two shards as threads on this GPU) reproduces the whole-chunk result."""
    import threading
    import torch
    from image_processing_amd import sharding, synth
    from image_processing_amd.engine import Engine
    from test_gpu_sharding import ThreadComm
    dev = torch.device("cuda", 0)
    n, world = 1000, 2
    fr = synth.chunk_torch(4, n, 2160, 3840, dev, speeds=[0, 0, 0, 4], segment=25)
    a = eng.detect_chunk_device(fr)
    b = eng.detect_chunk_device(fr)
    for k in ("flows", "keep", "counts", "cluster"):
        assert bool((a[k] == b[k]).all()), k
    kept = int(a["keep"].sum())
    assert 0 < kept < n and int(a["counts"][:, 0].sum()) > 0
    del b
    comm = ThreadComm(world)
    res, errors = [None] * world, []

    def work(rank):
        try:
            e = Engine(0)
            lo_, hi_ = sharding.shard_range(n, rank, world)
            lo, hi = sharding.halo_range(n, lo_, hi_, int(e.params.sliding_window_radius))
            out = sharding.detect_chunk_sharded(e, fr[lo:hi], lo_, hi_, n, comm.view(rank))
            torch.cuda.synchronize()
            res[rank] = {k: out[k].clone() for k in ("flows", "keep", "counts", "cluster")}
            e.close()
        except Exception as ex:                     # noqa: BLE001
            errors.append(ex)
            comm.barrier.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(400)
    assert not errors, errors
    for r in range(world):
        for k in ("flows", "keep", "counts", "cluster"):
            assert bool((res[r][k].to(a[k].dtype) == a[k]).all()), (r, k)

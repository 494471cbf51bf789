"""Frame-range sharding (image_processing_amd/sharding.py) over gloo, world sizes 2, 3 and 8, on the CPU:
the sharded result must equal the single-process pipeline bit for bit, including persistence runs
and static-map sums that straddle shard boundaries.  The compute stages are the oracle's
(tests/oracle_engine.py); what is under test is the host logic and the exchanges."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(seed, n, h, w, params):
    from image_processing_amd import synth
    frames = synth.chunk_numpy(seed, n, h, w, hot=10.0, stuck=0, motion_at=(5,), shift=6.0)
    # a pixel lit in frames 3..4 only and one lit in 7..9: runs that cross the shard boundaries
    frames[3:5, 4, 9] = 255
    frames[7:10, 20, 30] = 255
    frames[:, 11, 40] = 250          # stuck pixel: static / persistent
    return frames


def _worker(rank, world, port, seed, n, h, w, params, q):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from image_processing_amd import sharding
    from oracle_engine import OracleEngine
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        frames = _case(seed, n, h, w, params)
        eng = OracleEngine(params)
        a, b = sharding.shard_range(n, rank, world)
        lo, hi = sharding.halo_range(n, a, b, eng.params.sliding_window_radius)
        ext = torch.from_numpy(frames[lo:hi].copy())
        out = sharding.detect_chunk_sharded(eng, ext, a, b, n, sharding.make_comm())
        res = sharding.results_to_host(out, h * w)
        q.put((rank, {k: np.asarray(v) for k, v in res.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,params", [
    (2, 12, None),
    (2, 13, dict(consecutive_threshold=3, sliding_window_radius=2, static_threshold=20)),
    (3, 12, dict(consecutive_threshold=4, sliding_window_radius=4)),
    (2, 12, dict(consecutive_threshold=1)),
    (8, 41, dict(consecutive_threshold=3, sliding_window_radius=2)),     # the node's eight ranks, uneven ranges of 5 and 6 frames
])
def test_sharded_equals_single_process(world, n, params):
    from oracle import pipeline
    h, w, seed = 40, 72, 31
    frames = _case(seed, n, h, w, params)
    total, cc, sz, br = pipeline.detect_damaged_pixels(frames, params)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, n, h, w, params, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, res = q.get(timeout=240)
        got[r] = res
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):                               # every rank holds the global result
        res = got[r]
        assert np.array_equal(res["total_counts"], total, equal_nan=True), (r, res["total_counts"], total)
        assert np.array_equal(res["cluster_counts"], cc, equal_nan=True)
        assert np.array_equal(res["avg_sizes"], sz, equal_nan=True)
        assert np.array_equal(res["avg_brightnesses"], br, equal_nan=True)
    assert np.isnan(total).any()
    if not params or params.get("consecutive_threshold", 2) > 1:
        assert np.nansum(total) > 0          # (T = 1 makes every flagged pixel persistent: all zero)


def test_shard_and_halo_ranges():
    from image_processing_amd import sharding
    for n in (1, 7, 12, 1000):
        for world in (1, 2, 3, 8):
            if n < world:
                continue
            edges = [sharding.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            for a, b in edges:
                lo, hi = sharding.halo_range(n, a, b, 3)
                assert lo == max(0, a - 3) and hi == min(n, b + 3)
                assert sharding.halo_range(n, a, b, 0) == (max(0, a - 1), min(n, b + 1))

"""Frame-range sharding with the real HIP stages: the shards of one chunk run as threads on one
MI355X (one libipx context each) and exchange through an in-process stand-in for the process
group; the result must equal ipx_detect_chunk on the whole chunk bit for bit.  (The RCCL path
itself uses the same code with torch.distributed; it runs in bench.py --gpus N.)"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class ThreadComm:
    """all_gather / all_reduce among `world` threads of one process."""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def view(self, rank):
        outer = self

        class _V:
            world = outer.world

            def __init__(self):
                self.rank = rank

            def all_gather(self, t):
                import torch
                torch.cuda.synchronize()
                outer.slots[rank] = t.clone()
                outer.barrier.wait()
                out = [s.clone() for s in outer.slots]
                outer.barrier.wait()
                return out

            def all_reduce_sum(self, t):
                parts = self.all_gather(t)
                t.copy_(sum(parts[1:], parts[0]))
                return t
        return _V()


@pytest.mark.parametrize("world,n,params", [
    (2, 20, None),
    (3, 19, dict(consecutive_threshold=3, sliding_window_radius=2, static_threshold=20)),
    (4, 16, dict(consecutive_threshold=4, sliding_window_radius=5)),
])
def test_sharded_threads_equal_whole_chunk(world, n, params):
    import torch
    from image_processing_amd import sharding, synth
    from image_processing_amd.engine import Engine
    h, w = 90, 150
    frames = synth.chunk_numpy(7, n, h, w, hot=14.0, stuck=3, motion_at=(6, 11), shift=6.0)
    frames[3:6, 4, 9] = 255
    frames[8:12, 20, 30] = 255
    whole = Engine(0, params)
    ref = whole.detect_chunk(frames, want_masks=True)
    whole.close()
    comm = ThreadComm(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            eng = Engine(0, params)
            a, b = sharding.shard_range(n, rank, world)
            lo, hi = sharding.halo_range(n, a, b, eng.params.sliding_window_radius)
            ext = eng.to_device(frames[lo:hi])
            out = sharding.detect_chunk_sharded(eng, ext, a, b, n, comm.view(rank), want_final_bits=True)
            res = sharding.results_to_host(out, h * w)
            res["final"] = eng.unpack_mask(out["final_bits"], h, w).cpu().numpy().astype(bool)
            res["range"] = (a, b)
            results[rank] = res
            torch.cuda.synchronize()
            eng.close()
        except Exception as e:                      # noqa: BLE001
            errors.append(e)
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    for r in range(world):
        res = results[r]
        for k in ("total_counts", "cluster_counts", "avg_sizes", "avg_brightnesses", "flows", "keep"):
            assert np.array_equal(res[k], ref[k], equal_nan=True), (r, k)
        a, b = res["range"]
        for i in range(a, b):
            exp = ref["final_masks"][i]
            assert np.array_equal(res["final"][i - a], np.zeros((h, w), bool) if exp is None else exp)
    assert np.isnan(ref["total_counts"]).any() and np.nansum(ref["total_counts"]) > 0


def test_randomised_sharding_sweep():
    """tools/fuzz_sharded.py: random world sizes (2-4), chunk lengths down to the minimum a shard may hold,
    parameter sets and runs of lit pixels across shard edges; sharded == whole chunk, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_sharded
    from image_processing_amd.engine import Engine
    engines = [Engine(0) for _ in range(5)]
    try:
        for seed in range(3000, 3080):
            assert fuzz_sharded.one_case(engines, seed).startswith("ok")
    finally:
        for e in engines:
            e.close()

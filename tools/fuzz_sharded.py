#!/usr/bin/env python3
"""Randomised check of frame-range sharding: the shards of a random chunk run as threads on one GPU (one
libipx context each, an in-process stand-in for the process group) and must reproduce ipx_detect_chunk on
the whole chunk bit for bit (flows included), for random world sizes, chunk lengths and parameter sets.
    python tools/fuzz_sharded.py [cases] [seed] [--group]
--group: the same cases through ipx_group_detect_chunk instead (the sharding done inside libipx, all ranks on GPU 0,
peer-copy transport)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one_case(engines, seed):
    import torch
    from image_processing_amd import sharding, synth
    from test_gpu_sharding import ThreadComm
    rng = np.random.default_rng(seed)
    world = int(rng.integers(2, 5))
    T = int(rng.integers(1, 5)); radius = int(rng.integers(1, 6))
    n = int(rng.integers(world * max(T - 1, 1), 36))
    h = int(rng.integers(34, 120)); w = int(rng.integers(34, 200))
    params = dict(consecutive_threshold=T, sliding_window_radius=radius, brightness_threshold=int(rng.integers(120, 220)),
                  flow_threshold=float(rng.choice([0.5, 2.0, 4.0])), static_threshold=float(rng.choice([5, 20, 50])),
                  max_cluster_size=int(rng.integers(6, 40)))
    motion = tuple(int(x) for x in rng.choice(np.arange(1, n), size=min(int(rng.integers(0, 4)), n - 1), replace=False)) if n > 2 else ()
    frames = synth.chunk_numpy(seed, n, h, w, hot=float(rng.uniform(2, 30)), stuck=int(rng.integers(0, 6)),
                               motion_at=motion, shift=float(rng.uniform(0.5, 9)), blobs=int(rng.integers(0, 4)))
    for k in range(int(rng.integers(0, 4))):                 # pixels lit for a few consecutive frames: runs across shard edges
        s = int(rng.integers(0, n)); frames[s:s + int(rng.integers(1, 6)), int(rng.integers(0, h)), int(rng.integers(0, w))] = 255
    whole = engines[0]
    whole.set_params(params)
    fd = whole.to_device(frames)
    ref = {k: v.clone() for k, v in whole.detect_chunk_device(fd).items() if k in ("flows", "keep", "counts", "cluster")}
    torch.cuda.synchronize()
    comm = ThreadComm(world)
    res, errors = [None] * world, []

    def work(rank):
        try:
            e = engines[1 + rank]
            e.set_params(params)
            a, b = sharding.shard_range(n, rank, world)
            lo, hi = sharding.halo_range(n, a, b, radius)
            out = sharding.detect_chunk_sharded(e, fd[lo:hi].contiguous(), a, b, n, comm.view(rank))
            torch.cuda.synchronize()
            res[rank] = {k: out[k].clone() for k in ("flows", "keep", "counts", "cluster")}
        except Exception as ex:                     # noqa: BLE001
            errors.append(ex)
            comm.barrier.abort()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    for r in range(world):
        for k in ("flows", "keep", "counts", "cluster"):
            assert bool((res[r][k].to(ref[k].dtype) == ref[k]).all()), "rank %d %s" % (r, k)
    return "ok  world=%d n=%-2d %3dx%-3d T=%d r=%d kept=%d counts=%d" % (world, n, w, h, T, radius, int(ref["keep"].sum()), int(ref["counts"][:, 0].sum()))


def group_case(engine, groups, seed):
    """One random case through the C-level group (ranks = contexts on GPU 0)."""
    from image_processing_amd import synth
    from image_processing_amd.engine import Group
    rng = np.random.default_rng(seed)
    world = int(rng.integers(1, 7))
    T = int(rng.integers(1, 5)); radius = int(rng.integers(1, 9))
    n = int(rng.integers(max(world * max(T - 1, 1), 2), 48))
    h = int(rng.integers(34, 120)); w = int(rng.integers(34, 200))
    params = dict(consecutive_threshold=T, sliding_window_radius=radius, brightness_threshold=int(rng.integers(120, 220)),
                  flow_threshold=float(rng.choice([0.5, 2.0, 4.0])), static_threshold=float(rng.choice([5, 20, 50])),
                  max_cluster_size=int(rng.integers(6, 40)))
    motion = tuple(int(x) for x in rng.choice(np.arange(1, n), size=min(int(rng.integers(0, 4)), n - 1), replace=False)) if n > 2 else ()
    frames = synth.chunk_numpy(seed, n, h, w, hot=float(rng.uniform(2, 30)), stuck=int(rng.integers(0, 6)),
                               motion_at=motion, shift=float(rng.uniform(0.5, 9)), blobs=int(rng.integers(0, 4)))
    for k in range(int(rng.integers(0, 4))):
        s = int(rng.integers(0, n)); frames[s:s + int(rng.integers(1, 6)), int(rng.integers(0, h)), int(rng.integers(0, w))] = 255
    engine.set_params(params)
    ref = engine.detect_chunk(frames)
    if world not in groups:
        groups[world] = Group([0] * world, transport="peer")
    g = groups[world]
    g.set_params(params)
    out = g.detect_chunk(frames)
    for k in ("flows", "keep", "counts", "cluster", "total_counts", "cluster_counts", "avg_sizes", "avg_brightnesses"):
        assert np.array_equal(np.asarray(out[k]), np.asarray(ref[k]), equal_nan=True), k
    return "ok  group world=%d n=%-2d %3dx%-3d T=%d r=%d kept=%d counts=%d" % (world, n, w, h, T, radius, int(ref["keep"].sum()), int(ref["counts"][:, 0].sum()))


def main():
    from image_processing_amd.engine import Engine
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    group = "--group" in sys.argv
    cases = int(args[0]) if len(args) > 0 else 40
    seed0 = int(args[1]) if len(args) > 1 else 500
    engines = [Engine(0) for _ in range(1 if group else 5)]
    groups = {}
    t0 = time.time()
    for k in range(cases):
        try:
            msg = group_case(engines[0], groups, seed0 + k) if group else one_case(engines, seed0 + k)
        except AssertionError as e:
            print("seed %d: MISMATCH %s" % (seed0 + k, e), flush=True)
            sys.exit(1)
        print("seed %d: %s  [%.0f s]" % (seed0 + k, msg, time.time() - t0), flush=True)
    print("%d cases passed" % cases)


if __name__ == "__main__":
    main()

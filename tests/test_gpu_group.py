"""ipx_group_* (include/ipx.h): the frame-range sharding of one chunk INSIDE libipx -- one host thread, context and
stream per rank -- must equal ipx_detect_chunk on the whole chunk bit for bit.  On the one-GPU box the ranks of the
peer-copy transport all sit on device 0 (the sharding logic, halos, edge planes, OR / sum merges are the same code as
with distinct devices); the RCCL transport is exercised as a world of one, which still sends every exchange through
ncclAllGather / ncclAllReduce."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ("total_counts", "cluster_counts", "avg_sizes", "avg_brightnesses", "keep", "counts", "cluster")


def _frames(seed, n, h, w):
    from image_processing_amd import synth
    fr = synth.chunk_numpy(seed, n, h, w, hot=14.0, stuck=3, motion_at=(6, 11), shift=6.0)
    fr[3:6, 4, 9] = 255
    fr[8:12, 20, 30] = 255
    fr[:, 40:44, 60:64] = 250          # a static bright block: exercises the chunk-wide static map
    return fr


def _same(a, b):
    for k in KEYS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True), k
    assert np.array_equal(a["flows"], b["flows"]), "flows"


@pytest.mark.parametrize("world,n,params", [
    (1, 12, None),
    (2, 20, None),
    (3, 19, dict(consecutive_threshold=3, sliding_window_radius=2, static_threshold=20)),
    (4, 16, dict(consecutive_threshold=4, sliding_window_radius=5)),
    (2, 9, dict(consecutive_threshold=1)),
    (5, 41, dict(consecutive_threshold=2, sliding_window_radius=1, flow_threshold=0.4)),
])
def test_group_peer_equals_whole_chunk(world, n, params):
    from image_processing_amd.engine import Engine, Group
    fr = _frames(7 + world, n, 90, 150)
    ref = Engine(0, params).detect_chunk(fr)
    g = Group([0] * world, params, transport="peer")
    assert g.transport == "peer"
    _same(g.detect_chunk(fr), ref)
    _same(g.detect_chunk(fr), ref)                     # a second call on the same buffers
    fr2 = _frames(70 + world, n + 3, 64, 100)          # then another geometry through the same group
    _same(g.detect_chunk(fr2), Engine(0, params).detect_chunk(fr2))
    g.close()


def test_group_rccl_world_of_one():
    from image_processing_amd.engine import Engine, Group
    params = dict(consecutive_threshold=3, sliding_window_radius=2)
    fr = _frames(3, 18, 90, 150)
    ref = Engine(0, params).detect_chunk(fr)
    g = Group([0], params, transport="rccl")
    assert g.transport == "rccl"
    _same(g.detect_chunk(fr), ref)
    g.close()


def test_group_auto_and_errors():
    from image_processing_amd import _ipx
    from image_processing_amd.engine import Group
    g = Group([0], transport="auto")
    assert g.transport == "peer"                       # one device: nothing to exchange, RCCL is not loaded
    g.close()
    with pytest.raises(_ipx.IpxError):                 # RCCL refuses a device listed twice
        Group([0, 0], transport="rccl")
    with pytest.raises(_ipx.IpxError):
        Group([99])
    g = Group([0, 0, 0], dict(consecutive_threshold=4), transport="peer")
    with pytest.raises(_ipx.IpxError, match="every rank needs"):
        g.detect_chunk(_frames(1, 8, 32, 48))          # 8 frames over 3 ranks: fewer than T-1 = 3 per rank
    g.close()


@pytest.mark.parametrize("world,transport", [(1, "peer"), (3, "peer"), (1, "rccl")])
def test_group_two_chunks_in_flight(world, transport):
    """ipx_group_submit / ipx_group_wait: chunk k+1 is submitted (its frames uploaded on the ranks' side streams into
    their second frame buffer) while chunk k is being computed; every chunk must equal its own single-context result,
    also when geometry, length and PARAMETERS change from one chunk to the next, from pinned and from pageable memory."""
    import torch
    from image_processing_amd import _ipx
    from image_processing_amd.engine import Engine, Group
    pa, pb = dict(consecutive_threshold=3, sliding_window_radius=2), dict(consecutive_threshold=2, static_threshold=20)
    chunks = [(_frames(11, 18, 90, 150), pa), (_frames(12, 18, 90, 150), pa), (_frames(13, 23, 64, 100), pb),
              (_frames(14, 12, 90, 150), pb), (_frames(15, 18, 90, 150), pa)]
    refs = [Engine(0, p).detect_chunk(fr) for fr, p in chunks]
    g = Group([0] * world, pa, transport=transport)
    with pytest.raises(RuntimeError):
        g.wait()                                           # nothing in flight
    pinned = [torch.from_numpy(fr).pin_memory() for fr, _ in chunks]
    pending = []
    for k, (fr, p) in enumerate(chunks):
        g.set_params(p)                                    # applies to the chunks submitted from now on
        pending.append((k, g.submit(pinned[k].numpy() if k % 2 == 0 else fr)))
        if len(pending) == 2:
            if k == 1:
                with pytest.raises(_ipx.IpxError, match="two chunks are in flight"):
                    g.submit(fr)
                with pytest.raises(RuntimeError, match="in the order"):
                    g.wait(pending[1][1])
            j, h = pending.pop(0)
            _same(g.wait(h), refs[j])
    j, h = pending.pop(0)
    _same(g.wait(h), refs[j])
    _same(g.detect_chunk(chunks[0][0]), Engine(0, chunks[-1][1]).detect_chunk(chunks[0][0]))   # the synchronous form after it
    g.submit(chunks[1][0])
    g.close()                                              # closing with a chunk in flight waits for it


def test_group_full_size_shards():
    """1080p, 96 frames over 4 ranks on device 0: the sizes of the real workload (one strip per 114 columns, 128 x 64
    detect tiles, 17 + 9 words per row) through the group path."""
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine, Group
    import torch
    fr = synth.chunk_torch(5, 96, 1080, 1920, torch.device("cuda", 0)).cpu().numpy()
    ref = Engine(0).detect_chunk(fr)
    g = Group([0, 0, 0, 0], transport="peer")
    _same(g.detect_chunk(fr), ref)
    g.close()


def test_group_randomised_sweep():
    """tools/fuzz_sharded.py --group: random world sizes (1-6), chunk lengths down to the minimum a rank may hold,
    window radii up to 8, persistence thresholds 1-4 (3000 cases passed on the GPU box; 60 here)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_sharded
    from image_processing_amd.engine import Engine
    eng, groups = Engine(0), {}
    try:
        for seed in range(9000, 9060):
            assert fuzz_sharded.group_case(eng, groups, seed).startswith("ok")
    finally:
        for g in groups.values():
            g.close()

"""The RCCL path of image_processing_amd/sharding.py on the one-GPU box: a process group of ONE rank with
backend "nccl" (= RCCL on ROCm), with the exchanges of detect_chunk_sharded forced on, so that exactly the
tensors a multi-GPU run sends -- int64 bit-plane stacks (all_gather), the int32 heat / valid maps
(all_reduce sum), the packed int64 per-frame scalars (all_gather) -- go through RCCL's device collectives
before the first 8-GPU run does.  The result must equal the exchange-free path and ipx_detect_chunk bit for
bit.  The group lives in a fresh child process (forked from the launcher's fork server, which was started
before this process touched the GPU: conftest.py), so a problem inside RCCL cannot wedge the test session.

No scaling number comes from this: one rank measures no link."""
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rccl_world1(port, conn):
    """Runs in the child; sends ("ok", report) or ("error", traceback)."""
    import traceback
    try:
        import numpy as np
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        try:
            from image_processing_amd import sharding, synth
            from image_processing_amd.engine import Engine
            report = {}
            for name, params, n in (("defaults", None, 24),
                                    ("T3_r2", dict(consecutive_threshold=3, sliding_window_radius=2, static_threshold=20), 19),
                                    ("T1", dict(consecutive_threshold=1), 12)):
                h, w = 90, 150
                frames = synth.chunk_numpy(11, n, h, w, hot=14.0, stuck=3, motion_at=(6, 13), shift=6.0)
                frames[3:6, 4, 9] = 255
                eng = Engine(0, params)
                fr = eng.to_device(frames)
                comm = sharding.make_comm(force=True)
                assert isinstance(comm, sharding._Dist) and comm.exchange and not comm.stage
                assert dist.get_backend() == "nccl"
                comm.timed = True
                got = sharding.detect_chunk_sharded(eng, fr, 0, n, n, comm, want_final_bits=True)
                calls, nbytes = comm.calls, comm.bytes
                ms = comm.collective_ms()
                solo = sharding.detect_chunk_sharded(eng, fr, 0, n, n, sharding._Solo(), want_final_bits=True)
                whole = eng.detect_chunk_device(fr, want_final_bits=True)
                for k in ("flows", "keep", "counts", "cluster", "final_bits"):
                    assert bool((got[k] == solo[k]).all()), (name, k, "forced RCCL exchange != exchange-free path")
                    assert bool((got[k].reshape(whole[k].shape) == whole[k]).all()), (name, k, "sharded != ipx_detect_chunk")
                # the collectives themselves, on the dtypes and shapes used above
                planes = got["final_bits"][:2].contiguous()
                g = comm.all_gather(planes)
                assert len(g) == 1 and g[0].dtype == torch.int64 and bool((g[0] == planes).all())
                maps = torch.arange(2 * h * w, dtype=torch.int32, device=dev).reshape(2, h, w)
                r = comm.all_reduce_sum(maps.clone())
                assert r.dtype == torch.int32 and bool((r == maps).all())
                res = sharding.results_to_host(got, h * w)
                assert np.isnan(res["total_counts"]).any()
                assert name == "T1" or np.nansum(res["total_counts"]) > 0     # T = 1: every flagged pixel is persistent
                report[name] = dict(collectives=calls, bytes=nbytes, ms=round(ms, 3))
                eng.close()
            conn.send(("ok", report))
        finally:
            dist.destroy_process_group()
    except BaseException:                               # noqa: BLE001
        conn.send(("error", traceback.format_exc()))
    finally:
        conn.close()


def test_sharded_exchanges_over_rccl_world_of_one():
    import multiprocessing as mp
    from image_processing_amd import multi
    multi.ensure_launcher()                             # no-op: conftest.py started it before any GPU use
    ctx = mp.get_context("forkserver")
    parent, child = ctx.Pipe(duplex=False)
    p = ctx.Process(target=_rccl_world1, args=(_free_port(), child), daemon=True)
    p.start()
    child.close()
    try:
        assert parent.poll(240), "the RCCL child did not answer within 240 s"
        kind, payload = parent.recv()
    finally:
        p.join(20)
        if p.is_alive():
            p.terminate()
            p.join(10)
    assert kind == "ok", payload
    # 3 collectives for T <= 1 (persistence planes, maps, scalars), 4 otherwise (+ the edge planes)
    assert payload["defaults"]["collectives"] == 4 and payload["T3_r2"]["collectives"] == 4 and payload["T1"]["collectives"] == 3
    print("RCCL world-of-one collectives:", payload)

"""Several GPUs from the entry points a user actually starts, rehearsed on the one-GPU box:

* ``image_processing.main(video, gpus=...)`` deals the independent chunks of a video (ip.py:1204-1227) over the listed
  devices -- here device 0 listed twice: two pipelines, each with its own decode thread, libipx context and upload
  stream -- and must return what the single-device run returns, bit for bit; with fewer chunks than devices every chunk
  is sharded by frame range inside libipx (ipx_group_detect_chunk) instead;
* ``python bench.py --gpus 2`` started plainly, without a launcher: it must start its two ranks itself as fresh child
  processes (torch.distributed.run), relay ONE JSON line with n_gpus == 2, the four collectives per chunk and the
  strong-scaling figure.  IPX_BENCH_REHEARSE=1 puts both ranks on GPU 0 over gloo."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERIES = ("averages_counts", "averages_clusters", "averages_size", "averages_brightness", "times")


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a, float), np.asarray(b, float), equal_nan=True)


@pytest.fixture(scope="module")
def video(tmp_path_factory):
    from image_processing_amd import avi, synth
    frames = synth.chunk_numpy(77, 26, 72, 100, hot=8.0, stuck=2, motion_at=(5, 16), shift=6.0)
    fn = str(tmp_path_factory.mktemp("gpus") / "clip.avi")
    avi.write_avi(fn, list(frames), fps=4.0)
    return fn


def test_main_deals_chunks_over_devices(video, monkeypatch, capsys):
    import warnings
    import image_processing as ip
    kw = dict(average_time=1.0, max_chunks=None, step_size=7)          # 26 frames: chunks of 7, 7, 7, 5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = ip.main(video, **kw)
        capsys.readouterr()
        two = ip.main(video, gpus=[0, 0], **kw)
        out = capsys.readouterr().out
        assert sorted(l for l in out.splitlines() if l.startswith("processing chunk")) == \
            ["processing chunk %d: frames %d–%d" % (i, 7 * i, min(26, 7 * i + 7)) for i in range(4)]
        three = ip.main(video, gpus=[0, 0, 0], **kw)                    # uneven deal: 2 + 1 + 1
        monkeypatch.setenv("IPX_GPUS", "0,0")
        env = ip.main(video, **kw)
        monkeypatch.setenv("IPX_GPUS", "1")
        env1 = ip.main(video, **kw)
        monkeypatch.delenv("IPX_GPUS")
        dev0 = ip.main(video, gpus=[0], **kw)
        # fewer chunks than devices: the chunk is cut by frame range over the devices inside libipx
        short1 = ip.main(video, average_time=1.0, max_chunks=1, step_size=13)
        short2 = ip.main(video, average_time=1.0, max_chunks=1, step_size=13, gpus=[0, 0, 0])
        # a chunk too short to cut goes to the first device whole
        tiny1 = ip.main(video, average_time=1.0, max_chunks=1, step_size=3)
        tiny2 = ip.main(video, average_time=1.0, max_chunks=1, step_size=3, gpus=[0, 0])
    for k in SERIES:
        for other in (two, three, env, env1, dev0):
            assert _eq_nan(one[k], other[k]), k
        assert _eq_nan(short1[k], short2[k]), k
        assert _eq_nan(tiny1[k], tiny2[k]), k
    assert np.isnan(np.asarray(one["averages_counts"], float)).sum() < len(one["averages_counts"])
    with pytest.raises(ValueError):
        ip.main(video, gpus=64, **kw)
    # the command line takes the same thing
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cli = ip._cli(["--video_filename", video, "--step_size", "7", "--max_chunks", "none", "--gpus", "0,0",
                       "--average_plots", "False"])
    for k in SERIES:
        assert _eq_nan(one[k], cli[k]), k


def test_main_with_devices_resumes_from_checkpoints(video, tmp_path):
    import warnings
    import image_processing as ip
    ck = str(tmp_path / "ck")
    kw = dict(average_time=1.0, max_chunks=None, step_size=7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one = ip.main(video, **kw)
        first = ip.main(video, gpus=[0, 0], checkpoint_folder=ck, **kw)
        names = sorted(os.listdir(ck))
        assert len(names) == 4
        os.remove(os.path.join(ck, names[1]))                           # one chunk left to do on two devices: the group path
        again = ip.main(video, gpus=[0, 0], checkpoint_folder=ck, **kw)
    assert sorted(os.listdir(ck)) == names
    for k in SERIES:
        assert _eq_nan(one[k], first[k]) and _eq_nan(one[k], again[k]), k


def _run_plain_bench(cmd, env, conn):
    """In a fresh child of the fork server (no GPU state): start bench.py the way the driver does."""
    import subprocess
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
        conn.send((r.returncode, r.stdout, r.stderr[-4000:]))
    except BaseException as e:                          # noqa: BLE001
        conn.send((-1, "", repr(e)))
    finally:
        conn.close()


def test_bench_launches_its_own_ranks():
    import multiprocessing as mp
    from image_processing_amd import multi
    multi.ensure_launcher()                             # no-op: conftest.py started it before any GPU use
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IPX_BENCH_REHEARSE"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "96",
           "--width", "160", "--frames", "16", "--strong-frames", "20", "--profile-steps", "1"]
    ctx = mp.get_context("forkserver")
    parent, child = ctx.Pipe(duplex=False)
    p = ctx.Process(target=_run_plain_bench, args=(cmd, env, child), daemon=True)
    p.start()
    child.close()
    try:
        assert parent.poll(300), "bench.py --gpus 2 did not finish within 300 s"
        rc, out, err = parent.recv()
    finally:
        p.join(20)
        if p.is_alive():
            p.terminate()
            p.join(10)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                         # ONE line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["frames_per_gpu"] == 16 and "32-frame chunks" in d["config"]["workload"]
    assert d["collectives"]["calls_per_chunk"] == 4 and d["collectives"]["bytes_per_chunk"] > 0
    assert d["strong_1000"]["frames_per_gpu"] == 10 and d["strong_1000"]["scaling"] == "strong" and d["strong_1000"]["value"] > 0
    assert d["roofline"]["kernel"] and d["cpu_baseline"] is None

"""BASELINE.json configs[4], the multi-video batch (image_processing_amd/multi.py): videos are independent
units of work (ip.py:1204-1227), one worker process per GPU runs the chunk driver on its videos, nothing is
exchanged.  On the one-GPU box the workers share GPU 0 (a device listed twice): two replicas as two fresh
processes, each of which must return exactly what a single run of image_processing.main returns."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SERIES = ("averages_counts", "averages_clusters", "averages_size", "averages_brightness", "times")


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a, float), np.asarray(b, float), equal_nan=True)


@pytest.fixture(scope="module")
def videos(tmp_path_factory):
    from image_processing_amd import avi, synth
    d = tmp_path_factory.mktemp("multi")
    out = []
    for v, (n, h, w, mot) in enumerate(((23, 72, 100, (5, 14)), (17, 64, 88, (9,)), (11, 50, 70, ()))):
        frames = synth.chunk_numpy(40 + v, n, h, w, hot=8.0, stuck=2, motion_at=mot, shift=6.0)
        fn = str(d / ("video%d.avi" % v))
        avi.write_avi(fn, list(frames), fps=4.0)
        out.append(fn)
    return out


def test_two_replicas_on_one_gpu_equal_their_single_runs(videos, tmp_path):
    import image_processing as ip
    from image_processing_amd import multi
    kw = dict(average_time=1.0, max_chunks=None, step_size=6, params={"consecutive_threshold": 2, "sliding_window_radius": 2})
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        singles = [ip.main(v, **kw) for v in videos]
    # two worker processes, both on GPU 0, one video each
    res = multi.run_videos(videos[:2], devices=[0, 0], timeout=300, **kw)
    assert [r["video"] for r in res] == videos[:2] and all(r["device"] == 0 for r in res)
    for r, s in zip(res, singles):
        for k in SERIES:
            assert _eq_nan(r[k], s[k]), (r["video"], k)
    # three videos on two workers: worker 0 takes videos 0 and 2 one after the other; per-video checkpoint folders
    ck = str(tmp_path / "ck")
    res3 = multi.run_videos(videos, devices=[0, 0], timeout=300, checkpoint_folder=ck, **kw)
    for r, s in zip(res3, singles):
        for k in SERIES:
            assert _eq_nan(r[k], s[k]), (r["video"], k)
    assert sorted(os.listdir(ck)) == ["video0", "video1", "video2"]
    assert len(os.listdir(os.path.join(ck, "video0"))) == 4            # 23 frames in chunks of 6


def test_worker_failure_is_reported(videos, tmp_path):
    from image_processing_amd import multi
    with pytest.raises(RuntimeError) as ei:
        multi.run_videos([videos[0], str(tmp_path / "missing.avi")], devices=[0, 0], timeout=300, max_chunks=None, step_size=6)
    assert "worker 1 failed" in str(ei.value)


def test_command_line(videos, capsys):
    from image_processing_amd import multi
    res = multi._cli(["--videos", ",".join(videos[:2]), "--devices", "0,0", "--timeout", "300",
                      "--step_size", "6", "--max_chunks", "none", "--average_plots", "False"])
    out = capsys.readouterr().out
    assert len(res) == 2 and "Done: 2 videos" in out and "averages_counts" in out

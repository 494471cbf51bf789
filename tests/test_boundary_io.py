"""The input-side functions the drop-in keeps for API parity (download helpers, load_video_frames,
get_video_frames_from_url), restated from the reference's own tests: t.py = /root/reference/test_image_processing.py
lines 102-133 (download with a fake response), 150-165 (frame range from a fake capture) and 168-191 (stubbed callees).
t.py:159 replaces ``cv2.VideoCapture`` itself, which does not exist in this image: the fake capture is injected through
the module's ``_open_capture`` instead (same object protocol: isOpened / read / set / get / release)."""
import os

import numpy as np
import pytest

import image_processing as ip

POS_FRAMES = 1                                          # cv2.CAP_PROP_POS_FRAMES


class FakeResponse:                                     # t.py:32-37
    def __init__(self, status_code, content=b""):
        self.status_code = status_code
        self.content = content


class FakeVideo:                                        # t.py:40-75
    def __init__(self, frames, start=0, fail_after=None):
        self.frames = frames
        self.idx = start
        self.total_frames = len(frames)
        self.fail_after = fail_after if fail_after is not None else self.total_frames
        self.released = False

    def isOpened(self):
        return self.idx < self.fail_after and self.idx < self.total_frames

    def read(self):
        if not self.isOpened():
            return False, None
        frame = self.frames[self.idx]
        self.idx += 1
        return True, frame

    def set(self, prop, val):
        if prop == POS_FRAMES:
            self.idx = val

    def release(self):
        self.released = True

    def get(self, prop_id):
        return 0.0


def test_download_success(tmp_path, monkeypatch, capsys):                   # t.py:102-118
    import requests
    seen = []
    monkeypatch.setattr(requests, "get", lambda url: (seen.append(url), FakeResponse(200, b"video-content"))[1])
    filename = tmp_path / "out.mp4"
    result = ip.download_video_from_url("http://example.com/video.mp4", str(filename))
    assert result == str(filename) and seen == ["http://example.com/video.mp4"]
    with open(filename, "rb") as f:
        assert f.read() == b"video-content"
    assert f"Downloaded video as {filename}" in capsys.readouterr().out


def test_download_failure(tmp_path, monkeypatch, capsys):                   # t.py:121-133
    import requests
    monkeypatch.setattr(requests, "get", lambda url: FakeResponse(404))
    filename = tmp_path / "out.mp4"
    result = ip.download_video_from_url("http://example.com/video.mp4", str(filename))
    assert result == str(filename)
    assert not os.path.exists(str(filename))
    assert "Failed to download video" in capsys.readouterr().out


def _grey_frames(n=10):
    return [np.full((10, 10), 100 + i, dtype=np.uint8) for i in range(n)]


def test_load_selected_frames(monkeypatch):                                 # t.py:150-165 (grey frames: no GPU needed)
    caps = []

    def open_capture(filename):
        caps.append(FakeVideo(_grey_frames(), fail_after=10))
        return caps[-1], POS_FRAMES
    monkeypatch.setattr(ip, "_open_capture", open_capture)
    frames = ip.load_video_frames("dummy_frames", frames_start=2, frames_end=5)
    assert len(frames) == 3 and [int(f[0, 0]) for f in frames] == [102, 103, 104]      # frames 2, 3, 4
    assert all(f.ndim == 2 and f.dtype == np.uint8 for f in frames) and caps[-1].released
    # the whole video; a capture that fails after 6 frames; an end beyond the video; start without end
    assert len(ip.load_video_frames("dummy_frames")) == 10
    monkeypatch.setattr(ip, "_open_capture", lambda fn: (FakeVideo(_grey_frames(), fail_after=6), POS_FRAMES))
    assert len(ip.load_video_frames("dummy_frames")) == 6
    assert len(ip.load_video_frames("dummy_frames", frames_start=4, frames_end=50)) == 2
    assert [int(f[0, 0]) for f in ip.load_video_frames("dummy_frames", frames_start=3)] == [103, 104, 105]
    # frames_start = 0 / None do not seek (ip.py:65-66 tests truthiness)
    sets = []

    class Recording(FakeVideo):
        def set(self, prop, val):
            sets.append((prop, val))
            FakeVideo.set(self, prop, val)
    monkeypatch.setattr(ip, "_open_capture", lambda fn: (Recording(_grey_frames()), POS_FRAMES))
    ip.load_video_frames("dummy_frames", frames_start=0, frames_end=2)
    assert sets == []
    ip.load_video_frames("dummy_frames", frames_start=7)
    assert sets == [(POS_FRAMES, 7)]
    # grayscale=False hands the decoded frames over as they are (ip.py:78)
    colour = [np.full((4, 4, 3), 9, dtype=np.uint8)] * 3
    monkeypatch.setattr(ip, "_open_capture", lambda fn: (FakeVideo(colour), POS_FRAMES))
    out = ip.load_video_frames("dummy_frames", grayscale=False)
    assert len(out) == 3 and out[0].shape == (4, 4, 3)


@pytest.mark.gpu
def test_load_colour_frames_become_grey(monkeypatch):                       # t.py:136-165: colour frames -> 2-D greyscale
    dummy = [np.full((10, 10, 3), 128, dtype=np.uint8) for _ in range(10)]
    monkeypatch.setattr(ip, "_open_capture", lambda fn: (FakeVideo(dummy, fail_after=10), POS_FRAMES))
    frames = ip.load_video_frames("dummy_frames", frames_start=2, frames_end=5)
    assert len(frames) == 3 and all(f.ndim == 2 and f.dtype == np.uint8 for f in frames)
    assert all((f == 128).all() for f in frames)        # the luma weights sum to one: grey-coded colour keeps its value
    assert len(ip.load_video_frames("dummy_frames")) == 10


def test_get_video_frames_from_url(monkeypatch):                            # t.py:168-191
    called = {}

    def fake_download(url, filename):
        called["dl"] = (url, filename)
        return filename

    def fake_load(filename, start, end):
        called["ld"] = (filename, start, end)
        return ["frame1", "frame2"]
    monkeypatch.setattr(ip, "download_video_from_url", fake_download)
    monkeypatch.setattr(ip, "load_video_frames", fake_load)
    result = ip.get_video_frames_from_url("http://example.com/video", local_filename="my.avi", frames_start=1, frames_end=3)
    assert result == ["frame1", "frame2"]
    assert called["dl"] == ("http://example.com/video", "my.avi")
    assert called["ld"] == ("my.avi", 1, 3)
    # defaults (ip.py:137)
    ip.get_video_frames_from_url("http://example.com/video")
    assert called["dl"] == ("http://example.com/video", "temp_video.avi") and called["ld"] == ("temp_video.avi", None, None)

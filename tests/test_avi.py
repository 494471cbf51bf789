"""Uncompressed-AVI writer / reader round trip (the cv2-free input side), CPU only."""
import numpy as np
import pytest

from image_processing_amd import avi


@pytest.mark.parametrize("shape", [(6, 8), (5, 7), (9, 13, 3), (4, 6, 3)])
def test_roundtrip(tmp_path, shape):
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, shape).astype(np.uint8) for _ in range(5)]
    fn = str(tmp_path / "v.avi")
    avi.write_avi(fn, frames, fps=30.0)
    r = avi.AviReader(fn)
    assert r.frame_count == 5 and r.fps == pytest.approx(30.0)
    assert (r.height, r.width) == shape[:2]
    for i in (4, 0, 2):
        assert np.array_equal(r.read(i), frames[i])
    r.close()


def test_videocapture_subset(tmp_path):
    frames = [np.full((4, 6), i, np.uint8) for i in range(7)]
    fn = str(tmp_path / "v.avi")
    avi.write_avi(fn, frames, fps=25)
    cap = avi.VideoCapture(fn)
    assert cap.isOpened() and cap.get(cap.CAP_PROP_FRAME_COUNT) == 7 and cap.get(cap.CAP_PROP_FPS) == 25
    cap.set(cap.CAP_PROP_POS_FRAMES, 5)
    ok, f = cap.read()
    assert ok and f.shape == (4, 6, 3) and (f == 5).all()          # grey-coded BGR, as cv2 returns it
    assert cap.read()[0] and not cap.read()[0]
    cap.release()
    bad = avi.VideoCapture(str(tmp_path / "missing.avi"))
    assert not bad.isOpened() and bad.read() == (False, None)


def test_rejects_compressed(tmp_path):
    fn = str(tmp_path / "v.avi")
    avi.write_avi(fn, [np.zeros((4, 4), np.uint8)])
    data = bytearray(open(fn, "rb").read())
    i = data.find(b"strf") + 8 + 16
    data[i:i + 4] = b"MJPG"
    open(fn, "wb").write(bytes(data))
    with pytest.raises(ValueError, match="uncompressed"):
        avi.AviReader(fn)


def test_read_gray_returns_the_stored_frames(tmp_path):
    import numpy as np
    from image_processing_amd import avi
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, (13, 21), dtype=np.uint8) for _ in range(4)]
    fn = str(tmp_path / "g.avi")
    avi.write_avi(fn, frames, fps=10.0)
    cap = avi.VideoCapture(fn)
    assert cap.is_gray()
    cap.set(avi.VideoCapture.CAP_PROP_POS_FRAMES, 1)
    got = []
    while True:
        ok, f = cap.read_gray()
        if not ok:
            break
        got.append(f)
    cap.release()
    assert len(got) == 3 and all(np.array_equal(a, b) for a, b in zip(got, frames[1:]))
    avi.write_avi(fn, [np.repeat(f[:, :, None], 3, axis=2) for f in frames], fps=10.0)
    cap = avi.VideoCapture(fn)
    assert not cap.is_gray()
    cap.release()


def test_chunk_loader_short_cut_equals_load_video_frames(tmp_path, monkeypatch):
    """image_processing._load_chunk reads 8-bit AVIs straight into one buffer; the frames must be the ones
    load_video_frames returns (padded rows, bottom-up storage, ranges past the end), and a caller that
    replaces load_video_frames gets theirs."""
    import image_processing as ip
    rng = np.random.default_rng(3)
    fr = rng.integers(0, 256, (23, 37, 50), dtype=np.uint8)          # 50 columns: rows padded to 52 bytes
    fn = str(tmp_path / "g.avi")
    avi.write_avi(fn, fr)
    for start, end in ((0, 10), (5, 23), (20, 1000)):
        frames, _ = ip._load_chunk(fn, start, end)
        ref = ip.load_video_frames(fn, frames_start=start, frames_end=end)
        assert len(frames) == len(ref) == min(end, 23) - start
        assert all(np.array_equal(a, b) for a, b in zip(frames, ref))
        assert np.array_equal(frames[0], fr[start])
    assert ip._load_chunk_gray_avi(fn, 0, 10) is not None
    col = str(tmp_path / "c.avi")
    avi.write_avi(col, rng.integers(0, 256, (4, 8, 8, 3), dtype=np.uint8))
    assert ip._load_chunk_gray_avi(col, 0, 4) is None                  # colour: the general path
    monkeypatch.setattr(ip, "load_video_frames", lambda *a, **k: [np.zeros((2, 2), np.uint8)] * 3)
    assert ip._load_chunk_gray_avi(fn, 0, 10) is None
    frames, _ = ip._load_chunk(fn, 0, 10)
    assert len(frames) == 3 and frames[0].shape == (2, 2)


def test_write_avi_refuses_more_than_4_gb():
    class Fake(np.ndarray):
        pass
    big = np.zeros((1, 1), np.uint8)
    frames = [np.broadcast_to(big, (4096, 4096))] * 300                 # 300 x 16 MB > 4 GB, no memory used
    with pytest.raises(ValueError, match="4 GB"):
        avi.write_avi("/nonexistent/never_written.avi", frames)


def test_palette_of_8bit_streams_is_applied(tmp_path):
    """An 8-bit DIB stores palette indices.  Only the identity grey ramp makes them grey levels (the fast path);
    any other colour table is applied, as cv2 / FFmpeg do before BGR2GRAY."""
    frames = [np.arange(24, dtype=np.uint8).reshape(4, 6) * 10 + i for i in range(3)]
    fn = str(tmp_path / "ramp.avi")
    avi.write_avi(fn, frames, fps=25)
    r = avi.AviReader(fn)
    assert r.grey_palette and r.palette.shape == (256, 3) and (r.palette[77] == 77).all()
    r.close()
    cap = avi.VideoCapture(fn)
    assert cap.is_gray()
    cap.release()
    # the same file with an inverted, tinted colour table
    data = bytearray(open(fn, "rb").read())
    at = data.index(b"strf") + 8 + 40                       # colour table follows the 40-byte BITMAPINFOHEADER
    for i in range(256):
        data[at + 4 * i:at + 4 * i + 4] = bytes((255 - i, i // 2, i, 0))     # B, G, R, reserved
    fn2 = str(tmp_path / "tinted.avi")
    open(fn2, "wb").write(bytes(data))
    r = avi.AviReader(fn2)
    assert not r.grey_palette and r.palette[10].tolist() == [245, 5, 10]
    assert np.array_equal(r.read(1), frames[1])             # raw indices stay available
    r.close()
    cap = avi.VideoCapture(fn2)
    assert not cap.is_gray()
    ok, f = cap.read()
    assert ok and f.shape == (4, 6, 3)
    assert np.array_equal(f[..., 0], 255 - frames[0]) and np.array_equal(f[..., 1], frames[0] // 2) and np.array_equal(f[..., 2], frames[0])
    cap.release()


def test_failed_open_leaves_nothing_open(tmp_path):
    fn = str(tmp_path / "junk.avi")
    open(fn, "wb").write(b"RIFF" + b"\0" * 40)
    with pytest.raises(ValueError):
        avi.AviReader(fn)
    import gc
    gc.collect()
    import os
    os.remove(fn)                                           # (would fail on platforms that lock mapped files)
    assert not avi.VideoCapture(fn).isOpened()


def test_record_layout_and_region_read(tmp_path):
    """The chunk driver's raw path: frames [a, b) of a plain uncompressed AVI are one evenly spaced region of the file,
    read with large positional reads; record i of the region holds frame a + i as a bottom-up DIB with padded rows."""
    rng = np.random.default_rng(5)
    for shape in ((9, 13), (8, 16), (5, 7, 3)):
        frames = rng.integers(0, 256, (11,) + shape, dtype=np.uint8)
        fn = str(tmp_path / ("r%d.avi" % len(shape)))
        avi.write_avi(fn, list(frames), fps=10.0)
        r = avi.AviReader(fn)
        try:
            assert r.record_layout(0, 12) is None and r.record_layout(3, 3) is None
            for a, b in ((0, 11), (2, 7), (10, 11)):
                off0, rec, nbytes = r.record_layout(a, b)
                h, w = shape[:2]
                stride = r.stride
                assert stride % 4 == 0 and stride >= w * (3 if len(shape) == 3 else 1) and r.bottom_up
                assert nbytes == (b - a - 1) * rec + stride * h and (b - a == 1 or rec == stride * h + 8)
                buf = np.zeros(nbytes + 5, np.uint8)
                r.read_region_into(off0, buf[:nbytes], threads=3, piece=64)
                assert (buf[nbytes:] == 0).all()
                for i in range(b - a):
                    dib = buf[i * rec:i * rec + stride * h].reshape(h, stride)[::-1]
                    got = dib[:, :w] if len(shape) == 2 else dib[:, :3 * w].reshape(h, w, 3)
                    assert np.array_equal(got, frames[a + i])
        finally:
            r.close()

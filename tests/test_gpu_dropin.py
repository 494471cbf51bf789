"""The drop-in module (`import image_processing as ip`) on the GPU: the known answers of the
reference's own test file (t.py = /root/reference/test_image_processing.py, under the signatures
of the image_processing.py in the reference tree), the fused chunk path against the stage-by-stage
path and the oracle, the monkeypatch contract, and the chunk driver on an AVI file.
Needs an MI355X: ``-m gpu``."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ip():
    import image_processing
    return image_processing


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a, float), np.asarray(b, float), equal_nan=True)


def test_settings(ip):                                              # t.py:196-213
    s = ip._prepare_settings(None)
    assert (s.consecutive_threshold, s.brightness_threshold, s.flow_threshold, s.static_threshold) == (2, 170, 2.0, 50)
    assert (s.min_cluster_size, s.max_cluster_size, s.min_circularity, s.sliding_window_radius, s.number_of_plots) == \
        (5, 20, 0.1, 3, 20)
    s = ip._prepare_settings({"brightness_threshold": 99, "extra": 1})
    assert s.brightness_threshold == 99 and s.extra == 1 and s.flow_threshold == 2.0
    assert isinstance(s, type)


def test_compute_background_toy(ip):                                # SURVEY section 4 values of ip.compute_background
    toy = np.array([[[1, 5], [5, 1]], [[2, 6], [6, 2]], [[200, 200], [200, 200]],
                    [[7, 5], [3, 8]], [[3, 1], [4, 5]]], dtype=float)
    exp = [[[2, 6], [6, 2]], [[100.5, 102.5], [102.5, 100.5]], [[4.5, 5.5], [4.5, 5.0]],
           [[101.5, 100.5], [102.0, 102.5]], [[7, 5], [3, 8]]]
    for i in range(5):
        bg = ip.compute_background(toy, i, 1)
        assert bg.dtype == np.float32 and np.array_equal(bg, np.array(exp[i], np.float32))
    assert np.array_equal(ip.compute_background(toy, radius=1), np.array(exp, np.float32))   # newer batch form
    with pytest.raises(ValueError):
        ip.compute_background(toy + 0.25, 1, 1)


def test_find_background_matches_oracle(ip):
    from oracle import cref
    rng = np.random.default_rng(3)
    for n in (1, 3, 6):
        st = rng.integers(0, 256, (n, 17, 70)).astype(np.float32)
        for coeff in (1.0, 0.5):
            assert np.array_equal(ip._find_background(st, coeff), cref.find_background_f32(st, coeff))


def test_persistent_mask(ip):                                       # t.py:333-355
    m1 = np.array([[True, False], [True, True]])
    m2 = np.array([[True, True], [False, True]])
    p = ip.compute_persistent_mask([m1, m2], 2)
    assert p.dtype == bool and np.array_equal(p, [[True, False], [False, True]])
    m1 = np.array([[True, True], [False, True]])
    m3 = np.array([[True, True], [True, False]])
    for middle in (np.zeros((2, 2), bool), None):
        assert not ip.compute_persistent_mask([m1, middle, m3], 2).any()


def test_filter_consecutive_pixels(ip):                             # t.py:358-385
    masks = [np.array([[True, False], [True, True]]), np.array([[True, True], [False, True]])]
    pers = np.array([[True, False], [False, True]])
    f, c = ip.filter_consecutive_pixels(masks, pers)
    assert c == [1, 1] and np.array_equal(f[0], masks[0] & ~pers) and f[0].dtype == bool
    f, c = ip.filter_consecutive_pixels([None, np.array([[True]])], np.array([[False]]))
    assert f[0] is None and np.isnan(c[0]) and c[1] == 1
    assert len(ip.filter_consecutive_pixels(masks, pers, return_counts=False)) == 2


def test_final_count(ip):                                           # t.py:422-430
    masks = [np.array([[1, 1], [0, 1]], bool), np.zeros((2, 2), bool)]
    assert ip._get_final_count(masks, [10, np.nan]).tolist() == [13.0, 0.0]
    out = ip._get_final_count([masks[0], None], [10, np.nan])
    assert out[0] == 13.0 and np.isnan(out[1])


def test_static_mask_and_suppression(ip):                           # t.py:461-493
    frames = np.array([[[10, 200], [200, 10]], [[10, 200], [200, 10]], [[10, 50], [50, 10]]])
    masks = [f > 170 for f in frames]
    assert np.array_equal(ip.compute_static_mask(frames, masks, 180, 50, 1), [[False, True], [True, False]])
    m1 = np.array([[True, False], [True, True]])
    pers = np.array([[True, False], [False, False]])
    stat = np.array([[False, True], [False, False]])
    fm, counts = ip.apply_static_suppression([m1, np.zeros((2, 2), bool), None], pers, stat)
    exp = m1 & ~(pers | stat)
    assert np.array_equal(fm[0], exp) and counts[0] == exp.sum() and counts.dtype == np.float64
    assert not fm[1].any() and counts[1] == 0 and fm[2] is None and np.isnan(counts[2])


def test_damaged_pixel_mask(ip):                                    # t.py:559-582, both call forms
    frame = np.zeros((3, 3)); frame[1, 1] = 100
    for args in ((frame, np.zeros((3, 3))), (frame, 3, 3, np.zeros((3, 3)))):
        m, thr = ip.get_damaged_pixel_mask(*args)
        exp = np.zeros((3, 3), np.uint8); exp[1, 1] = 1
        assert m.dtype == np.uint8 and np.array_equal(m, exp) and np.all(thr == 30) and thr.shape == (3, 3)
    frame = np.zeros((2, 2)); frame[0, 0] = 200.0
    bg = np.zeros((2, 2)); bg[0, 0] = 255.0
    m, thr = ip.get_damaged_pixel_mask(frame, bg)
    assert thr[0, 0] == pytest.approx(255.0) and m[0, 0] == 0
    assert ip._raw_damaged_mask(np.zeros((3, 3)), np.zeros((3, 3))).dtype == bool


def test_filter_clusters(ip):                                       # t.py:585-615
    frame = np.arange(16).reshape(4, 4).astype(float)
    m = np.zeros((4, 4), np.uint8)
    m[0, 0] = m[0, 1] = 1
    m[3, 2] = m[3, 3] = m[2, 3] = 1
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)
    for extra in ((), (cross,)):
        cleaned, count, size, bright = ip.filter_damaged_pixel_clusters(frame, m, 2, 2, 0.0, *extra)
        assert count == 1 and cleaned.sum() == 2 and cleaned.dtype == bool and size == 2.0
        assert bright == (frame[0, 0] + frame[0, 1]) / 2
    cleaned, count, size, bright = ip.filter_damaged_pixel_clusters(frame, np.zeros((4, 4), bool), 2, 2, 0.0)
    assert count == 0 and size == 0.0 and np.isnan(bright) and not cleaned.any()


def test_remove_bright_regions(ip):                                 # t.py:618-646
    bg = np.array([[50, 60], [70, 80]], float)
    d = np.zeros((2, 2), bool); d[1, 1] = True
    assert np.array_equal(ip.remove_bright_regions(bg, 100, d, 2), d)
    bg = np.array([[200, 200, 10], [200, 200, 10], [10, 10, 10]])
    d = np.zeros((3, 3), bool); d[0, 0] = True; d[2, 2] = True
    exp = np.zeros((3, 3), bool); exp[2, 2] = True
    assert np.array_equal(ip.remove_bright_regions(bg, 100, d, 2), exp)


def test_estimates(ip):                                             # t.py:649-703
    frames = np.array([[[10, 10], [200, 200]], [[255, 255], [255, 255]]], np.float32)
    masks = np.array([[[True, False], [False, False]], [[False, False], [False, False]]])
    out = ip.estimate_damaged_pixels_in_bright_areas(frames, masks, 100)
    assert out[0] == pytest.approx(2.0) and np.isnan(out[1])
    out = ip.estimate_damaged_pixels_in_bright_areas([np.full((2, 2), 200.0)],
                                                     [np.array([[False, True], [False, False]])], 100)
    assert out.shape == (1,) and np.isnan(out[0])
    frames = np.array([[[150, 200], [50, 80]], [[10, 20], [30, 40]]])
    out = ip.find_bright_area_estimates(frames, list(np.zeros((2, 2, 2), bool)), 170)
    assert out[0] == pytest.approx(0) and np.isnan(out[1])


def test_flow_metric(ip):                                           # t.py:706-727
    z = np.zeros((8, 8), np.uint8)
    assert np.allclose(ip.compute_optical_flow_metric([z, z.copy(), z.copy()]), 0.0)
    f0 = np.zeros((10, 10)); f1 = f0.copy(); f0[3, 3] = 255; f1[7, 7] = 255
    fl = ip.compute_optical_flow_metric([f0, f1])
    assert fl.shape == (2,) and fl.dtype == np.float64 and fl[0] == 0.0 and fl[1] > 0.0


def test_filter_frames_and_heatmap(ip):                             # t.py:730-786
    frames = np.arange(3 * 4).reshape(3, 2, 2)
    ff, fc, fm, fl = ip.filter_frames_by_optical_flow(frames, np.array([1, 2, 3]), np.array([0.1, 5.0, 0.2]),
                                                      np.zeros((3, 2, 2), bool), 1.0)
    assert len(ff) == 2 and fc.tolist() == [1, 3] and fl.tolist() == [0.1, 0.2]
    frames = np.zeros((12, 2, 2), np.uint8)
    masks = np.zeros((12, 2, 2), bool); masks[:6, 0, 0] = True
    h = ip.find_damaged_pixel_heatmap(frames, masks, 170)
    assert h[0, 0] == pytest.approx(50.0) and h[1, 1] == 0.0
    assert ip.find_damaged_pixel_heatmap(frames[:5], masks[:5], 170).max() == 0.0     # <= 10 valid frames


def test_chunked_nanmean(ip):
    assert ip.chunked_nanmean(np.array([1, 2, np.nan, 4, 5.0]), 2) == [1.5, 4.0, 5.0]
    assert np.isnan(ip.chunked_nanmean(np.array([]), 3)[0])


def _chunk():
    from image_processing_amd import synth
    return synth.chunk_numpy(41, 18, 72, 120, motion_at=(6, 7), shift=6.0, hot=12.0, stuck=3)


def test_detect_fused_equals_oracle_and_stagewise(ip, monkeypatch):
    from oracle import pipeline
    frames = _chunk()
    for params in (None, dict(consecutive_threshold=3, brightness_threshold=150, static_threshold=20,
                              min_cluster_size=2, max_cluster_size=12, min_circularity=0.5, sliding_window_radius=2)):
        exp = pipeline.detect_damaged_pixels(frames, params)
        got = ip.detect_damaged_pixels(list(frames), 0, params=params)
        assert all(g.dtype == np.float64 and g.shape == (len(frames),) for g in got)
        assert _eq_nan(np.stack(got), np.stack(exp))
        # replace one stage by a counting wrapper: the module must go through its globals, stage by stage
        calls = {"n": 0}
        orig = ip.remove_bright_regions

        def wrapped(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        monkeypatch.setattr(ip, "remove_bright_regions", wrapped)
        staged = ip.detect_damaged_pixels(list(frames), 0, params=params)
        monkeypatch.undo()
        assert calls["n"] == int(np.sum(~np.isnan(exp[0])))
        assert _eq_nan(np.stack(staged), np.stack(exp))
    assert np.isnan(exp[0]).any() and np.nansum(exp[0]) > 0


def test_detect_respects_patched_flow(ip, monkeypatch):             # t.py:496-556 style: stub stages
    frames = _chunk()[:6]
    monkeypatch.setattr(ip, "compute_optical_flow_metric", lambda fr: np.array([0.0, 9.0, 0.0, 0.0, 9.0, 0.0]))
    total, cc, sz, br = ip.detect_damaged_pixels(frames, 0)
    assert np.isnan(total).tolist() == [False, True, False, False, True, False]
    assert cc[1] == 0 and sz[1] == 0 and br[1] == 0


def test_main_on_avi(ip, tmp_path):
    from image_processing_amd import avi
    from oracle import pipeline, stages as S
    frames = _chunk()
    fn = str(tmp_path / "clip.avi")
    avi.write_avi(fn, list(frames), fps=4.0)
    loaded = ip.load_video_frames(fn, 3, 9)
    assert len(loaded) == 6 and loaded[0].ndim == 2 and np.array_equal(np.stack(loaded), frames[3:9])
    res = ip.main(fn, average_time=1.0, max_chunks=None, step_size=7)
    series = [[], [], [], []]
    for a in range(0, len(frames), 7):
        for s, v in zip(series, pipeline.detect_damaged_pixels(frames[a:a + 7])):
            s.append(v)
    exp = [np.concatenate(s) for s in series]
    for key, e in zip(("averages_counts", "averages_clusters", "averages_size", "averages_brightness"), exp):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert _eq_nan(res[key], S.chunked_nanmean(e, 4)), key
    assert np.allclose(res["times"], (np.arange(len(res["averages_counts"])) * 4 + 2) / 4.0)
    # colour AVI: grey-coded BGR decodes to the same grey frames
    fn2 = str(tmp_path / "clip_bgr.avi")
    avi.write_avi(fn2, [np.repeat(f[:, :, None], 3, axis=2) for f in frames[:4]], fps=4.0)
    assert np.array_equal(np.stack(ip.load_video_frames(fn2)), frames[:4])
    # ... and the chunk driver, which uploads such a file's bytes as stored and converts on the GPU (ipx_unpack_dib),
    # returns what it returns for the greyscale file
    fn3 = str(tmp_path / "clip_bgr_all.avi")
    avi.write_avi(fn3, [np.repeat(f[:, :, None], 3, axis=2) for f in frames], fps=4.0)
    res3 = ip.main(fn3, average_time=1.0, max_chunks=None, step_size=7)
    for key in ("averages_counts", "averages_clusters", "averages_size", "averages_brightness", "times"):
        assert _eq_nan(res[key], res3[key]), key


def test_next_video_is_prefetched_and_changes_nothing(ip, tmp_path):
    """queue_next_video: the first chunk of the video after this one is decoded under this one's last chunk; main() on
    it then takes that chunk over.  Results are those of plain runs; a prefetch that does not fit is dropped."""
    from image_processing_amd import avi, synth
    files, plain = [], []
    for k in range(3):
        frames = synth.chunk_numpy(40 + k, (17, 18, 20)[k], 72, 96, motion_at=(5,), hot=8.0, stuck=2)
        fn = str(tmp_path / ("v%d.avi" % k))
        avi.write_avi(fn, list(frames), fps=5.0)
        files.append(fn)
        plain.append(ip.main(fn, average_time=1.0, max_chunks=None, step_size=6))
    before = dict(ip._PREFETCH_STATS)
    got = []
    for k, fn in enumerate(files):
        ip.queue_next_video(files[k + 1] if k + 1 < len(files) else None, 6)
        got.append(ip.main(fn, average_time=1.0, max_chunks=None, step_size=6))
    assert ip._PREFETCH_STATS["started"] - before["started"] == 2 and ip._PREFETCH_STATS["taken"] - before["taken"] == 2
    for a, b in zip(plain, got):
        for key in ("averages_counts", "averages_clusters", "averages_size", "averages_brightness", "times"):
            assert _eq_nan(a[key], b[key]), key
    # announced with one chunking, processed with another: the prefetched chunk is dropped, the result is the plain one
    ip.queue_next_video(files[1], 6)
    ip.main(files[0], average_time=1.0, max_chunks=None, step_size=6)
    taken = ip._PREFETCH_STATS["taken"]
    other = ip.main(files[1], average_time=1.0, max_chunks=None, step_size=9)
    assert ip._PREFETCH_STATS["taken"] == taken and not ip._PREFETCHED
    again = ip.main(files[1], average_time=1.0, max_chunks=None, step_size=9)
    for key in ("averages_counts", "averages_clusters", "averages_size", "averages_brightness", "times"):
        assert _eq_nan(other[key], again[key]), key


def test_main_streams_chunks_like_the_sequential_driver(ip, tmp_path, monkeypatch):
    """The double-buffered driver (decode thread + asynchronous chunks) returns exactly what chunk-by-chunk
    calls of detect_damaged_pixels return, also with a partial last chunk and with a replaced stage."""
    from image_processing_amd import avi, synth
    frames = synth.chunk_numpy(5, 23, 72, 96, motion_at=(9,), hot=8.0, stuck=2)
    fn = str(tmp_path / "clip.avi")
    avi.write_avi(fn, list(frames), fps=5.0)
    res = ip.main(fn, average_time=1.0, max_chunks=None, step_size=6)
    series = [[], [], [], []]
    for a in range(0, len(frames), 6):
        for s, v in zip(series, ip.detect_damaged_pixels(frames[a:a + 6], 0)):
            s.append(v)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for key, e in zip(("averages_counts", "averages_clusters", "averages_size", "averages_brightness"), series):
            assert _eq_nan(res[key], ip.chunked_nanmean(np.concatenate(e), 5)), key
        # a replaced stage sends every chunk through the module-global detect_damaged_pixels
        calls = []
        orig = ip.compute_optical_flow_metric
        monkeypatch.setattr(ip, "compute_optical_flow_metric", lambda fr: (calls.append(len(fr)), orig(fr))[1])
        res2 = ip.main(fn, average_time=1.0, max_chunks=2, step_size=6)
    assert calls == [6, 6]
    assert _eq_nan(res2["averages_counts"], ip.chunked_nanmean(np.concatenate(series[0][:2]), 5))


def test_main_checkpoints_and_resumes(ip, tmp_path, monkeypatch):
    from image_processing_amd import avi, synth
    frames = synth.chunk_numpy(9, 17, 64, 80, hot=6.0, stuck=2)
    fn = str(tmp_path / "clip.avi")
    avi.write_avi(fn, list(frames), fps=4.0)
    ck = str(tmp_path / "ck")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        first = ip.main(fn, max_chunks=None, step_size=5, checkpoint_folder=ck)
        import os
        names = sorted(os.listdir(ck))
        assert len(names) == 4 and all(n.endswith(".npz") for n in names)
        os.remove(os.path.join(ck, names[2]))                      # only this chunk has to be decoded again
        loaded = []
        orig = ip.load_video_frames
        monkeypatch.setattr(ip, "load_video_frames", lambda f, frames_start=None, frames_end=None, grayscale=True:
                            (loaded.append((frames_start, frames_end)), orig(f, frames_start, frames_end, grayscale))[1])
        again = ip.main(fn, max_chunks=None, step_size=5, checkpoint_folder=ck)
    assert loaded == [(10, 15)]
    for k in ("averages_counts", "averages_clusters", "averages_size", "averages_brightness", "times"):
        assert _eq_nan(first[k], again[k]), k
    # (not in the reference) the margin of the motion test over the chunks computed in the run: all four, then the one
    assert first["min_flow_margin"] > 0 and again["min_flow_margin"] >= first["min_flow_margin"]


def test_randomised_labelling_stages_sweep():
    """tools/fuzz_stages.py: filter_damaged_pixel_clusters and remove_bright_regions of this module against
    the oracle on random masks (noise of any density, lines, rings, solid and hollow shapes, border contact)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fuzz_stages
    import image_processing as ip
    for seed in range(50000, 50200):
        assert fuzz_stages.one_case(ip, seed).startswith("ok")
        assert fuzz_stages.temporal_case(ip, seed) == "ok"


def test_load_video_frames_colour_batches(tmp_path):
    """Colour footage is converted on the GPU in batches of 32 decoded frames: same grey frames as the luma
    restatement frame by frame, across batch boundaries and with a frame range."""
    import image_processing as ip
    from image_processing_amd import avi
    from oracle import cref
    rng = np.random.default_rng(12)
    bgr = rng.integers(0, 256, (70, 20, 26, 3), dtype=np.uint8)
    fn = str(tmp_path / "colour.avi")
    avi.write_avi(fn, bgr, fps=10.0)
    got = ip.load_video_frames(fn)
    assert len(got) == 70 and all(g.shape == (20, 26) and g.dtype == np.uint8 for g in got)
    assert np.array_equal(np.stack(got), np.stack([cref.bgr2gray_u8(f) for f in bgr]))
    part = ip.load_video_frames(fn, frames_start=30, frames_end=67)
    assert np.array_equal(np.stack(part), np.stack([cref.bgr2gray_u8(f) for f in bgr[30:67]]))
    raw = ip.load_video_frames(fn, grayscale=False)
    assert np.array_equal(np.stack(raw), bgr)


def test_load_video_frames_applies_a_non_grey_palette(ip, tmp_path):
    """An 8-bit AVI whose colour table is not the grey ramp: frames are the luma of the palette colours (what
    cv2.VideoCapture + cvtColor return), through load_video_frames and through main's chunk loader."""
    from image_processing_amd import avi
    from oracle import cref
    rng = np.random.default_rng(3)
    idx = rng.integers(0, 256, (9, 16, 24), dtype=np.uint8)
    fn = str(tmp_path / "ramp.avi")
    avi.write_avi(fn, idx, fps=4.0)
    data = bytearray(open(fn, "rb").read())
    at = data.index(b"strf") + 8 + 40
    pal = np.stack([255 - np.arange(256), np.arange(256) // 2, np.arange(256)], axis=1).astype(np.uint8)
    for i in range(256):
        data[at + 4 * i:at + 4 * i + 4] = bytes((int(pal[i, 0]), int(pal[i, 1]), int(pal[i, 2]), 0))
    fn2 = str(tmp_path / "tinted.avi")
    open(fn2, "wb").write(bytes(data))
    exp = np.stack([cref.bgr2gray_u8(pal[f]) for f in idx])
    assert np.array_equal(np.stack(ip.load_video_frames(fn2)), exp)
    assert np.array_equal(np.stack(ip.load_video_frames(fn)), idx)              # the grey ramp: stored bytes
    frames, pinned = ip._load_chunk(fn2, 2, 7)
    assert np.array_equal(np.stack(frames), exp[2:7])


def test_checkpoints_are_tied_to_video_and_settings(ip, tmp_path):
    """A checkpoint folder reused with other settings, or for another video of the same length, must not hand back
    the stale series; a corrupt file is recomputed and overwritten."""
    import os
    import warnings
    from image_processing_amd import avi, synth
    a = synth.chunk_numpy(21, 12, 48, 64, hot=8.0, stuck=2)
    b = synth.chunk_numpy(22, 12, 48, 64, hot=8.0, stuck=2)
    fa, fb = str(tmp_path / "a.avi"), str(tmp_path / "b.avi")
    avi.write_avi(fa, list(a), fps=4.0)
    avi.write_avi(fb, list(b), fps=4.0)
    ck = str(tmp_path / "ck")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kw = dict(max_chunks=None, step_size=6)
        ra = ip.main(fa, checkpoint_folder=ck, **kw)
        strict = ip.main(fa, checkpoint_folder=ck, params={"brightness_threshold": 100, "consecutive_threshold": 5}, **kw)
        fresh = ip.main(fa, params={"brightness_threshold": 100, "consecutive_threshold": 5}, **kw)
        assert _eq_nan(strict["averages_counts"], fresh["averages_counts"])     # recomputed, not the stored series
        assert not _eq_nan(strict["averages_counts"], ra["averages_counts"])
        rb = ip.main(fb, checkpoint_folder=ck, **kw)                             # another video, same length, same folder
        assert _eq_nan(rb["averages_counts"], ip.main(fb, **kw)["averages_counts"])
        # the folder now holds b's chunks: a is recomputed again and overwrites them
        names = sorted(os.listdir(ck))
        with open(os.path.join(ck, names[0]), "wb") as f:
            f.write(b"not an npz")
        again = ip.main(fa, checkpoint_folder=ck, **kw)
        for k in ra:
            assert _eq_nan(again[k], ra[k]), k
        with np.load(os.path.join(ck, names[0])) as z:                           # rewritten, readable, a's fingerprint
            assert "a.avi" in str(z["fingerprint"])

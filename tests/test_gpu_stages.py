"""HIP stages (through the C ABI of include/ipx.h) vs the CPU oracle, stage by stage.

Bit-exact for masks, planes, counters and the f32 background; the Farneback flow field is
compared with a stated tolerance (two reductions are formed in a different order on the GPU, csrc/flow.hip).
Needs an MI355X: run with ``-m gpu``.
"""
import numpy as np
import pytest

from conftest import arrays_to_masks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from image_processing_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def S():
    from oracle import stages
    return stages


def _bits_to_bool(eng, bits, H, W):
    return eng.unpack_mask(bits, H, W).cpu().numpy().astype(bool)


def _chunk(seed, n, h, w, **kw):
    from image_processing_amd import synth
    return synth.chunk_numpy(seed, n, h, w, **kw)


SHAPES = [(40, 56), (64, 64), (70, 131), (97, 200), (33, 257)]


def test_pack_unpack_roundtrip(eng):
    import torch
    rng = np.random.default_rng(0)
    for (h, w) in SHAPES + [(1, 1), (3, 64), (2, 65)]:
        m = rng.random((3, h, w)) < 0.4
        bits = eng.pack_mask(torch.from_numpy(m).to(eng.device))
        assert bits.shape == (3, h, (w + 63) // 64)
        back = _bits_to_bool(eng, bits, h, w)
        assert np.array_equal(back, m)
        # padding bits are zero
        words = bits.cpu().numpy().view(np.uint64)
        if w % 64:
            assert not (words[..., -1] >> np.uint64(w % 64)).any()


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("radius", [1, 3, 5])
def test_background_and_raw_planes(eng, S, shape, radius):
    h, w = shape
    n = 9
    frames = _chunk(5 + h + radius, n, h, w, hot=8.0)
    eng.set_params(dict(sliding_window_radius=radius))
    fd = eng.to_device(frames)
    raw, bright, dense = eng.detect_raw(fd)
    raw = _bits_to_bool(eng, raw, h, w)
    bright = _bits_to_bool(eng, bright, h, w)
    dense = dense.cpu().numpy()
    for i in range(n):
        bg = S.compute_background(frames, i, radius)
        got = eng.background(fd, i).cpu().numpy()
        assert got.dtype == np.float32 and np.array_equal(got, bg), "background frame %d" % i
        assert np.array_equal(raw[i], S.raw_damaged_mask(frames[i], bg)), "raw mask frame %d" % i
        assert np.array_equal(bright[i], bg > 170), "bright plane frame %d" % i
        assert dense[i, 0] == (frames[i] < 170).sum() and dense[i, 1] == (frames[i] > 170).sum()
    eng.set_params(None)


@pytest.mark.parametrize("radius", [1, 2, 3, 4, 6, 8])
def test_background_integer_decision_on_ties_and_extremes(eng, S, radius):
    """detect.hip decides _find_background's outlier cut in integers wherever that is provably the reference's
    decision and runs the float chain elsewhere.  Frames drawn from a few grey levels make exact ties
    (t^2 == D, dyadic and non-dyadic means), near-ties and zero variance common; 0 / 255 mixtures reach the
    largest D; chunk edges give every neighbour count from radius to 2 radius.  The f32 background must equal the
    oracle's float chain bit for bit, also for a coefficient other than 1 (which always takes the chain)."""
    rng = np.random.default_rng(100 + radius)
    h, w, n = 40, 136, 2 * radius + 4
    for levels in ([0, 255], [7, 8], [10, 11, 13], [0, 1, 2, 3], [100, 101, 102, 103, 104, 105, 250], list(range(256))):
        frames = rng.choice(np.array(levels, np.uint8), size=(n, h, w))
        frames[:, :4, :8] = levels[0]                                    # a flat patch: D = 0
        for coeff in (1.0, 0.5, 1.75):
            eng.set_params(dict(sliding_window_radius=radius, pixel_std_coeff=coeff))
            fd = eng.to_device(frames)
            for i in (0, 1, radius, n // 2, n - 2, n - 1):
                exp = S.compute_background(frames, i, radius, coeff)
                got = eng.background(fd, i).cpu().numpy()
                assert np.array_equal(got, exp), (levels[:4], coeff, i, int((got != exp).sum()))
            if coeff == 1.0:                                             # the planes of the fused kernel as well
                raw, bright, _ = eng.detect_raw(fd)
                raw = _bits_to_bool(eng, raw, h, w)
                for i in (0, n // 2, n - 1):
                    bg = S.compute_background(frames, i, radius)
                    assert np.array_equal(raw[i], S.raw_damaged_mask(frames[i], bg))
    eng.set_params(None)


def test_raw_planes_discarded_frames_are_zero(eng, S):
    import torch
    frames = _chunk(3, 8, 48, 80, hot=10.0)
    keep = np.array([1, 0, 1, 1, 0, 1, 1, 1], np.uint8)
    raw, bright, _ = eng.detect_raw(eng.to_device(frames), keep=torch.from_numpy(keep).to(eng.device))
    raw = _bits_to_bool(eng, raw, 48, 80)
    for i in range(8):
        if keep[i]:
            assert np.array_equal(raw[i], S.raw_damaged_mask(frames[i], S.compute_background(frames, i, 3)))
        else:
            assert not raw[i].any()


@pytest.mark.parametrize("f64", [False, True])
def test_damaged_mask_stage(eng, S, f64):
    rng = np.random.default_rng(2)
    for (h, w) in SHAPES + [(3, 3), (2, 2)]:
        frame = rng.integers(0, 256, (h, w)).astype(np.uint8)
        bg = (rng.random((h, w)) * 255).astype(np.float64 if f64 else np.float32)
        m, thr = S.get_damaged_pixel_mask(frame, bg)
        gm, gthr = eng.damaged_mask(eng.to_device(frame), eng.to_device(bg))
        assert np.array_equal(gthr.cpu().numpy(), thr)
        assert np.array_equal(gm.cpu().numpy(), m)


def test_damaged_mask_known_answers(eng):          # t.py:559-582
    frame = np.zeros((3, 3), np.uint8); frame[1, 1] = 100
    m, thr = eng.damaged_mask(eng.to_device(frame), eng.to_device(np.zeros((3, 3))))
    exp = np.zeros((3, 3), np.uint8); exp[1, 1] = 1
    assert np.all(thr.cpu().numpy() == 30) and np.array_equal(m.cpu().numpy(), exp)
    frame = np.zeros((2, 2), np.uint8); frame[0, 0] = 200
    bg = np.zeros((2, 2)); bg[0, 0] = 255.0
    m, thr = eng.damaged_mask(eng.to_device(frame), eng.to_device(bg))
    assert thr.cpu().numpy()[0, 0] == pytest.approx(255.0) and m.cpu().numpy()[0, 0] == 0


def _blobby(rng, h, w, k, maxpix):
    m = np.zeros((h, w), bool)
    for _ in range(k):
        y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
        for _ in range(int(rng.integers(1, maxpix))):
            m[min(h - 1, max(0, y + int(rng.integers(-2, 3)))), min(w - 1, max(0, x + int(rng.integers(-3, 4))))] = True
    return m


@pytest.mark.parametrize("shape", SHAPES + [(130, 300)])
def test_remove_bright_regions(eng, S, shape):
    import torch
    h, w = shape
    rng = np.random.default_rng(h * w)
    cases = []
    for dens_b, dens_d in [(0.02, 0.05), (0.3, 0.1), (0.55, 0.3), (0.9, 0.5)]:
        bright = rng.random((h, w)) < dens_b
        bright |= _blobby(rng, h, w, 6, 60)
        dmg = (rng.random((h, w)) < dens_d) | _blobby(rng, h, w, 10, 25)
        cases.append((bright, dmg))
    cases.append((np.ones((h, w), bool), rng.random((h, w)) < 0.2))      # one blob covering everything
    cases.append((np.zeros((h, w), bool), rng.random((h, w)) < 0.2))
    snake = np.zeros((h, w), bool); snake[::2, :] = True; snake[1::4, -1] = True; snake[3::4, 0] = True
    cases.append((snake, rng.random((h, w)) < 0.1))                      # long serpentine component
    bb = eng.pack_mask(torch.from_numpy(np.stack([c[0] for c in cases])).to(eng.device))
    db = eng.pack_mask(torch.from_numpy(np.stack([c[1] for c in cases])).to(eng.device))
    out = _bits_to_bool(eng, eng.remove_bright_regions(bb, db, h, w), h, w)
    for i, (bright, dmg) in enumerate(cases):
        bg = np.where(bright, 200.0, 10.0).astype(np.float32)
        exp = S.remove_bright_regions(bg, 170, dmg, 20)
        assert np.array_equal(out[i], exp), "case %d" % i


def test_remove_bright_regions_known_answer(eng):   # t.py:618-646
    import torch
    bright = np.array([[1, 1, 0], [1, 1, 0], [0, 0, 0]], bool)
    d = np.zeros((3, 3), bool); d[0, 0] = True; d[2, 2] = True
    eng.set_params(dict(max_cluster_size=2))
    out = eng.remove_bright_regions(eng.pack_mask(torch.from_numpy(bright[None]).to(eng.device)),
                                    eng.pack_mask(torch.from_numpy(d[None]).to(eng.device)), 3, 3)
    exp = np.zeros((3, 3), bool); exp[2, 2] = True
    assert np.array_equal(_bits_to_bool(eng, out, 3, 3)[0], exp)
    eng.set_params(None)


def test_temporal_stages_vs_reference_golden(eng, golden):
    """persistence / static map / suppression against the reference's own outputs."""
    import torch
    g = golden
    frames, kept = g["tmp_frames"], g["tmp_kept"]
    n, h, w = frames.shape
    fd = eng.to_device(frames)
    kd = eng.to_device(kept)
    raw_bits = eng.pack_mask(torch.from_numpy(g["tmp_raw"]).to(eng.device))
    for T in (1, 2, 3, 5):
        eng.set_params(dict(consecutive_threshold=T))
        p = _bits_to_bool(eng, eng.persistent_mask(raw_bits, kd, h, w)[None], h, w)[0]
        assert np.array_equal(p, g["persist_T%d" % T]), "T=%d" % T
    for mv in (10, 3):
        for stt in (50, 20, 5):
            eng.set_params(dict(consecutive_threshold=3, static_threshold=stt, min_valid_frames=mv))
            pers = eng.persistent_mask(raw_bits, kd, h, w)
            maps = eng.static_maps(fd, raw_bits, pers, kd)
            st = _bits_to_bool(eng, eng.static_from_maps(maps)[None], h, w)[0]
            assert np.array_equal(st, g["static_mv%d_st%d" % (mv, stt)])
    eng.set_params(dict(consecutive_threshold=3, static_threshold=5, min_valid_frames=3))
    pers = eng.persistent_mask(raw_bits, kd, h, w)
    st = eng.static_from_maps(eng.static_maps(fd, raw_bits, pers, kd))
    fin, counts = eng.finalize(fd, raw_bits, pers, st, kd)
    fin = _bits_to_bool(eng, fin, h, w)
    assert np.array_equal(fin, g["tmp_final"])
    counts = counts.cpu().numpy()
    exp_counts = np.nan_to_num(g["tmp_final_counts"], nan=0.0)
    assert np.array_equal(counts[:, 0], exp_counts)
    eng.set_params(None)


@pytest.mark.parametrize("T", [0, 1, 2, 4, 7, 40, 300])
def test_persistent_mask_random(eng, S, T):
    import torch
    rng = np.random.default_rng(T)
    n, h, w = 45, 19, 150
    raw = rng.random((n, h, w)) < 0.6
    keep = (rng.random(n) < 0.85).astype(np.uint8)
    keep[0] = 1
    raw[~keep.astype(bool)] = False
    masks = arrays_to_masks(keep, raw)
    eng.set_params(dict(consecutive_threshold=T))
    got = eng.persistent_mask(eng.pack_mask(torch.from_numpy(raw).to(eng.device)), eng.to_device(keep), h, w)
    assert np.array_equal(_bits_to_bool(eng, got[None], h, w)[0], S.compute_persistent_mask(masks, T))
    eng.set_params(None)


def test_static_and_finalize_random(eng, S):
    import torch
    rng = np.random.default_rng(9)
    n, h, w = 30, 21, 131
    frames = rng.integers(0, 256, (n, h, w)).astype(np.uint8)
    raw = rng.random((n, h, w)) < 0.3
    raw[:, 5, 7] = True                              # a stuck pixel
    keep = (rng.random(n) < 0.8).astype(np.uint8); keep[0] = 1
    raw[~keep.astype(bool)] = False
    masks = arrays_to_masks(keep, raw)
    eng.set_params(dict(consecutive_threshold=6, static_threshold=25))
    fd, kd = eng.to_device(frames), eng.to_device(keep)
    rb = eng.pack_mask(torch.from_numpy(raw).to(eng.device))
    pers_b = eng.persistent_mask(rb, kd, h, w)
    pers = S.compute_persistent_mask(masks, 6)
    clean, _ = S.filter_consecutive_pixels(masks, pers)
    static = S.compute_static_mask(frames, clean, 170, 25, 10)
    maps = eng.static_maps(fd, rb, pers_b, kd)
    st_b = eng.static_from_maps(maps)
    assert np.array_equal(_bits_to_bool(eng, st_b[None], h, w)[0], static)
    final, _ = S.apply_static_suppression(clean, pers, static)
    fin_b, counts = eng.finalize(fd, rb, pers_b, st_b, kd)
    fin = _bits_to_bool(eng, fin_b, h, w)
    counts = counts.cpu().numpy()
    for i in range(n):
        if keep[i]:
            assert np.array_equal(fin[i], final[i])
            assert counts[i, 0] == final[i].sum()
            assert counts[i, 1] == (final[i] & (frames[i] < 170)).sum()
            assert counts[i, 2] == (final[i] & (frames[i] > 170)).sum()
        else:
            assert not fin[i].any() and not counts[i].any()
    eng.set_params(None)


def _cluster_expect(S, frame, mask, st):
    cm, cnt, sz, br = S.filter_damaged_pixel_clusters(frame, mask, st["min_cluster_size"], st["max_cluster_size"],
                                                      st["min_circularity"])
    return cm, cnt, sz, br


@pytest.mark.parametrize("params", [dict(), dict(min_cluster_size=2, max_cluster_size=30, min_circularity=0.6),
                                    dict(min_cluster_size=10, max_cluster_size=20, min_circularity=0.9),
                                    dict(min_cluster_size=1, max_cluster_size=60, min_circularity=0.45)])
def test_cluster_stats(eng, S, params):
    import torch
    rng = np.random.default_rng(17)
    st = S.settings(params)
    eng.set_params(params)
    for (h, w) in [(32, 48), (70, 131), (64, 256)]:
        masks = np.stack([_blobby(rng, h, w, 14, 30) for _ in range(5)] + [np.zeros((h, w), bool)])
        frames = rng.integers(0, 256, masks.shape).astype(np.uint8)
        stats, cleaned = eng.cluster_stats(eng.to_device(frames), eng.pack_mask(torch.from_numpy(masks).to(eng.device)),
                                           want_cleaned=True)
        stats = stats.cpu().numpy()
        cleaned = _bits_to_bool(eng, cleaned, h, w)
        for i in range(len(masks)):
            cm, cnt, sz, br = _cluster_expect(S, frames[i], masks[i], st)
            assert stats[i, 0] == cnt, (h, w, i)
            assert np.array_equal(cleaned[i], cm), (h, w, i)
            if cnt:
                assert stats[i, 1] / cnt == sz and stats[i, 2] / stats[i, 1] == br
    eng.set_params(None)


def test_cluster_stats_reference_golden(eng, golden):
    import torch
    g = golden
    h, w = g["cl_mask"].shape
    for tag, (mn, mx, mc) in dict(a=(5, 20, 0.1), b=(2, 30, 0.6), c=(10, 20, 0.9)).items():
        eng.set_params(dict(min_cluster_size=mn, max_cluster_size=mx, min_circularity=mc))
        stats, cleaned = eng.cluster_stats(eng.to_device(g["cl_frame"][None]),
                                           eng.pack_mask(torch.from_numpy(g["cl_mask"][None]).to(eng.device)), want_cleaned=True)
        stats = stats.cpu().numpy()[0]
        exp = g["cl_%s_stats" % tag]
        assert np.array_equal(_bits_to_bool(eng, cleaned, h, w)[0], g["cl_%s_mask" % tag])
        assert stats[0] == exp[0]
        if exp[0]:
            assert stats[1] / stats[0] == exp[1] and stats[2] / stats[1] == exp[2]
    eng.set_params(None)


def test_cluster_close_border_known_answer(eng):    # t.py:585-615
    import torch
    frame = np.arange(16).reshape(4, 4).astype(np.uint8)
    m = np.zeros((4, 4), bool)
    m[0, 0] = m[0, 1] = True
    m[3, 2] = m[3, 3] = m[2, 3] = True
    eng.set_params(dict(min_cluster_size=2, max_cluster_size=2, min_circularity=0.0))
    stats, cleaned = eng.cluster_stats(eng.to_device(frame[None]), eng.pack_mask(torch.from_numpy(m[None]).to(eng.device)),
                                       want_cleaned=True)
    assert stats.cpu().numpy()[0, 0] == 1 and _bits_to_bool(eng, cleaned, 4, 4)[0].sum() == 2
    eng.set_params(None)


def test_bgr2gray(eng):
    from oracle import cref
    rng = np.random.default_rng(4)
    bgr = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    got = eng.bgr2gray(eng.to_device(bgr)).cpu().numpy()
    assert np.array_equal(got, cref.bgr2gray_u8(bgr))
    grey = np.repeat(rng.integers(0, 256, (5, 7, 1)).astype(np.uint8), 3, axis=2)
    assert np.array_equal(eng.bgr2gray(eng.to_device(grey)).cpu().numpy(), grey[..., 0])   # identity on grey-coded video


@pytest.mark.parametrize("shape", [(9, 13), (8, 16), (31, 100), (5, 7, 3), (12, 64, 3)])
def test_unpack_dib_equals_the_avi_reader(eng, shape, tmp_path):
    """ipx_unpack_dib (the chunk driver's input step on the GPU: rows turned over, padding dropped, BGR -> grey) against
    the frames the AVI reader + luma restatement return, for any width (dword and byte paths), from frame ranges."""
    import torch
    from image_processing_amd import avi
    from oracle import cref
    rng = np.random.default_rng(sum(shape))
    frames = rng.integers(0, 256, (7,) + shape, dtype=np.uint8)
    fn = str(tmp_path / "d.avi")
    avi.write_avi(fn, list(frames), fps=5.0)
    r = avi.AviReader(fn)
    try:
        for a, b in ((0, 7), (2, 5), (6, 7)):
            off0, rec, nbytes = r.record_layout(a, b)
            buf = np.empty(nbytes, np.uint8)
            r.read_region_into(off0, buf, threads=2, piece=128)
            got = eng.unpack_dib(torch.from_numpy(buf).to(eng.device), rec, b - a, shape[0], shape[1], r.bpp, r.stride,
                                 r.bottom_up).cpu().numpy()
            exp = frames[a:b] if len(shape) == 2 else np.stack([cref.bgr2gray_u8(f) for f in frames[a:b]])
            assert np.array_equal(got, exp)
    finally:
        r.close()


@pytest.mark.parametrize("radius", [1, 3, 6])
def test_raw_planes_pretest_list_and_dense_tiles(eng, S, radius):
    """detect_kernel's round-3 structure on its aligned tiles: the min / max row pre-test, the work list of the groups that
    pass it, and the per-row loop for tiles whose list overflows -- on content that takes each path (dark noise with hot
    pixels: short lists; all-bright noise: every group flagged, the dense loop; half and half; 0 / 255 mixtures: ties and
    flags everywhere; flat frames: nothing flagged), at several brightness thresholds (fractional, 0, at and beyond 255,
    negative = pre-test off), window radii 1, 3 and 6 (the 16-neighbour kernel) and with discarded frames."""
    import torch
    rng = np.random.default_rng(900 + radius)
    n = 2 * radius + 5
    for (h, w) in ((70, 256), (97, 388), (40, 132)):
        kinds = []
        dark = np.clip(rng.normal(70, 3, (n, h, w)), 0, 255)
        hot = rng.random((n, h, w)) < 0.002
        dark[hot] = rng.integers(150, 256, int(hot.sum()))
        kinds.append(np.rint(dark).astype(np.uint8))
        kinds.append(rng.integers(175, 256, (n, h, w)).astype(np.uint8))
        half = kinds[0].copy()
        half[:, :, w // 2:] = kinds[1][:, :, w // 2:]
        kinds.append(half)
        kinds.append(rng.choice(np.array([0, 255], np.uint8), size=(n, h, w)))
        kinds.append(np.full((n, h, w), 90, np.uint8))
        for k, frames in enumerate(kinds):
            for thr in ((170, 100.5, 254) if k != 1 else (170, 0, 255, 300.0, -5.0)):
                keep = (rng.random(n) < 0.8).astype(np.uint8)
                keep[0] = 1
                eng.set_params(dict(sliding_window_radius=radius, brightness_threshold=thr))
                raw, bright, dense = eng.detect_raw(eng.to_device(frames), keep=torch.from_numpy(keep).to(eng.device))
                raw = _bits_to_bool(eng, raw, h, w)
                bright = _bits_to_bool(eng, bright, h, w)
                dense = dense.cpu().numpy()
                for i in range(n):
                    if not keep[i]:
                        assert not raw[i].any() and not bright[i].any()
                        continue
                    bg = S.compute_background(frames, i, radius)
                    assert np.array_equal(raw[i], S.raw_damaged_mask(frames[i], bg)), (h, w, k, thr, i, "raw")
                    assert np.array_equal(bright[i], bg > np.float32(thr)), (h, w, k, thr, i, "bright")
                    assert dense[i, 0] == (frames[i] < 170).sum() and dense[i, 1] == (frames[i] > thr).sum(), (h, w, k, thr, i)
    eng.set_params(None)

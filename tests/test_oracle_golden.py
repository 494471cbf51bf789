"""Oracle vs tests/golden/reference_run.npz (outputs of the reference's own Python, produced by
tests/golden/make_golden.py in the build container).  CPU only."""
import numpy as np
import pytest

from conftest import arrays_to_masks
from oracle import pipeline, stages as S


def _eq_nan(a, b):
    return np.array_equal(np.asarray(a, float), np.asarray(b, float), equal_nan=True)


def test_temporal_stages(golden):
    g = golden
    frames, kept = g["tmp_frames"], g["tmp_kept"]
    raw = arrays_to_masks(kept, g["tmp_raw"])
    for T in (1, 2, 3, 5):
        assert np.array_equal(S.compute_persistent_mask(raw, T), g[f"persist_T{T}"])
    pers = S.compute_persistent_mask(raw, 3)
    clean, cc = S.filter_consecutive_pixels(raw, pers)
    assert all((c is None) == (not k) for c, k in zip(clean, kept))
    assert np.array_equal(np.stack([np.zeros_like(pers) if c is None else c for c in clean]), g["tmp_clean"])
    assert _eq_nan(cc, g["tmp_clean_counts"])
    for mv in (10, 3):
        for stt in (50, 20, 5):
            assert np.array_equal(S.compute_static_mask(frames, clean, 170, stt, mv), g[f"static_mv{mv}_st{stt}"])
    static = S.compute_static_mask(frames, clean, 170, 5, 3)
    final, fc = S.apply_static_suppression(clean, pers, static)
    assert np.array_equal(np.stack([np.zeros_like(pers) if c is None else c for c in final]), g["tmp_final"])
    assert _eq_nan(fc, g["tmp_final_counts"])
    assert _eq_nan(S.estimate_damaged_pixels_in_bright_areas(frames.astype(np.float64), final), g["tmp_estimate"])
    for bt in (170, 100, 254):
        assert _eq_nan(S.find_bright_area_estimates(frames, final, bt), g[f"tmp_bright_est_bt{bt}"])
    assert _eq_nan(S.get_final_count(final, S.find_bright_area_estimates(frames, final, 170)), g["tmp_final_count"])


def test_chunked_nanmean(golden):
    for step in (2, 3, 8, 9):
        assert _eq_nan(S.chunked_nanmean(golden["nanmean_in"], step), golden[f"nanmean_step{step}"])


def test_background(golden):
    """The reference body run as plain Python accumulates in float32 (no numba here), the
    restatement follows numba's float64 typing: they must agree wherever the
    float32 accumulation is exact, and may differ by one ulp elsewhere (none on these inputs)."""
    f = golden["bg_frames"]
    for r in (1, 3):
        ours = np.stack([S.compute_background(f, i, r) for i in range(len(f))])
        ref = golden[f"bg_r{r}"]
        assert ours.dtype == np.float32
        diff = np.abs(ours.astype(np.float64) - ref.astype(np.float64))
        assert (diff <= np.spacing(np.abs(ref)).astype(np.float64)).all()
        assert (diff > 0).mean() < 0.01
    toy = np.array([[[1, 5], [5, 1]], [[2, 6], [6, 2]], [[200, 200], [200, 200]],
                    [[7, 5], [3, 8]], [[3, 1], [4, 5]]], dtype=float)
    assert np.array_equal(np.stack([S.compute_background(toy, i, 1) for i in range(5)]), golden["bg_toy_r1"])


def test_hybrid_stages(golden):
    g = golden
    ch = g["hy_frames"]
    assert np.array_equal(S.compute_optical_flow_metric(ch), g["hy_flows"])
    for i in range(len(ch)):
        bg = S.compute_background(ch, i, 3)
        assert np.array_equal(bg, g["hy_bg"][i])
        m = S.raw_damaged_mask(ch[i], bg)
        assert np.array_equal(m, g["hy_rawmask"][i])
        assert np.array_equal(S.remove_bright_regions(bg, 170, m, 20), g["hy_brightfiltered"][i])


def test_hybrid_pipeline(golden):
    g = golden
    ch = g["hy_frames"]
    res = np.stack(pipeline.detect_damaged_pixels(ch))
    assert _eq_nan(res, g["hy_detect"])
    alt = dict(consecutive_threshold=3, brightness_threshold=150, flow_threshold=0.05, static_threshold=20,
               min_cluster_size=2, max_cluster_size=12, min_circularity=0.5, sliding_window_radius=2)
    assert _eq_nan(np.stack(pipeline.detect_damaged_pixels(ch, alt)), g["hy_detect_alt"])


@pytest.mark.parametrize("tag,args", [("a", (5, 20, 0.1)), ("b", (2, 30, 0.6)), ("c", (10, 20, 0.9))])
def test_cluster_filter(golden, tag, args):
    cm, cnt, sz, br = S.filter_damaged_pixel_clusters(golden["cl_frame"], golden["cl_mask"], *args)
    assert np.array_equal(cm, golden[f"cl_{tag}_mask"])
    assert _eq_nan([cnt, sz, br], golden[f"cl_{tag}_stats"])

"""Boundary code of the drop-in module that needs no GPU: the call forms of the newer API generation
that the reference's own test file uses with every stage stubbed (t.py = /root/reference/
test_image_processing.py; the cases below restate t.py:388-419, 433-458 and 496-556 as fixtures), the
command line (ip.py:1250-1316), the plot functions' output files (ip.py:999, 1029, 1062, 1127) under the
Agg backend, and the checkpoint fingerprint."""
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def ip():
    import matplotlib
    matplotlib.use("Agg")
    import image_processing
    return image_processing


class DummySettings:                                                 # t.py:78-88 (kernel = cv2's 3x3 "ellipse" = a cross)
    consecutive_threshold = 2
    brightness_threshold = 170
    flow_threshold = 5
    static_threshold = 50
    min_cluster_size = 2
    max_cluster_size = 10
    min_circularity = 0.3
    sliding_window_radius = 3
    number_of_plots = 20
    kernel = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], np.uint8)


def test_cluster_stats_flow_skips_with_kernel_argument(ip, monkeypatch):       # t.py:388-403
    frames = [np.zeros((2, 2)) for _ in range(3)]
    masks = [np.ones((2, 2), bool) for _ in range(3)]
    seen = []
    monkeypatch.setattr(ip, "filter_damaged_pixel_clusters",
                        lambda frame, mask, min_cluster_size, max_cluster_size, min_circularity, kernel:
                        (seen.append(kernel), (None, 42, 7, 123))[1])
    cc, cs, cb = ip.compute_cluster_stats(frames, masks, [0.0, 6.0, 0.0], DummySettings)
    assert cc.tolist() == [42, 0, 42] and cs.tolist() == [7, 0, 7] and cb.tolist() == [123, 0, 123]
    assert len(seen) == 2 and all(k is DummySettings.kernel for k in seen)


def test_cluster_stats_all_false_mask_skips(ip, monkeypatch):                  # t.py:406-419
    frames = [np.zeros((2, 2)) for _ in range(2)]
    masks = [np.zeros((2, 2), bool), np.ones((2, 2), bool)]
    monkeypatch.setattr(ip, "filter_damaged_pixel_clusters",
                        lambda frame, mask, min_cluster_size, max_cluster_size, min_circularity, kernel: (None, 5, 2, 50))
    cc, cs, cb = ip.compute_cluster_stats(frames, masks, [0.0, 0.0], DummySettings)
    assert cc.tolist() == [0, 5] and cs.tolist() == [0, 2] and cb.tolist() == [0, 50]


def test_cluster_stats_patched_five_argument_form(ip, monkeypatch):            # this tree's ip.py:336 (no kernel)
    settings = type("S", (), dict(flow_threshold=2.0, min_cluster_size=5, max_cluster_size=20, min_circularity=0.1))
    monkeypatch.setattr(ip, "filter_damaged_pixel_clusters",
                        lambda frame, mask, mn, mx, circ: (None, 3, 6.5, np.nan))
    masks = [np.zeros((2, 2), bool), None, np.ones((2, 2), bool)]
    out = ip.compute_cluster_stats([np.zeros((2, 2))] * 3, masks, [0.0, 0.0, 9.0], settings)
    # an all-False mask is NOT None under this tree's API: the stage is called (ip.py:334-336)
    assert out[0].tolist() == [3, 0, 0] and out[1].tolist() == [6.5, 0, 0] and np.isnan(out[2][0]) and out[2][1:].tolist() == [0, 0]


def test_generate_plots_newer_form(ip, monkeypatch):                           # t.py:433-458
    frames = np.zeros((4, 1, 1))
    masks = np.ones((4, 1, 1), bool)
    counts = np.array([1, 2, 3, 4])
    flows = np.array([0.0, 1.0, 2.0, 3.0])
    settings = DummySettings()
    settings.flow_threshold = 2.0
    settings.number_of_plots = 2
    called = {"visualisation": [], "plot_counts": 0, "heatmap_input": None}
    monkeypatch.setattr(ip, "visualise_damaged_pixels",
                        lambda frame, i, mask, count, show_plots, save_plots, output_folder: called["visualisation"].append(i))
    monkeypatch.setattr(ip, "plot_damaged_pixels",
                        lambda counts, show_plots, save_plots, output_folder:
                        called.__setitem__("plot_counts", called["plot_counts"] + 1))
    monkeypatch.setattr(ip, "find_damaged_pixel_heatmap", lambda frame, mask, brightness_threshold: np.array([[9]]))
    monkeypatch.setattr(ip, "plot_heatmap",
                        lambda heatmap, show_plots, save_plots, output_folder: called.__setitem__("heatmap_input", heatmap))
    ip._generate_plots(frames, masks, counts, flows, settings)
    assert called["visualisation"] == [0, 1]                                   # only frames 0 and 1 are plotted
    assert called["plot_counts"] == 1
    assert np.array_equal(called["heatmap_input"], np.array([[9]]))


def test_generate_plots_this_trees_form(ip, monkeypatch):                      # ip.py:381-435
    called = []
    monkeypatch.setattr(ip, "visualise_damaged_pixels",
                        lambda frame, frame_index, chunk_index, cluster_mask, cluster_count, bright_threshold=170,
                        show_plots=True, save_plots=False, output_folder="results":
                        called.append(("vis", frame_index, chunk_index, output_folder)))
    monkeypatch.setattr(ip, "plot_damaged_pixels",
                        lambda counts, chunk_index, show_plots=True, save_plots=False, output_folder="results":
                        called.append(("counts", chunk_index, show_plots)))
    monkeypatch.setattr(ip, "find_damaged_pixel_heatmap", lambda frames, masks, bt: np.zeros((1, 1)))
    monkeypatch.setattr(ip, "plot_heatmap",
                        lambda heatmap, chunk_index, title="t", show_plots=True, save_plots=False, output_folder="results",
                        figsize=(1, 1): called.append(("heat", chunk_index, save_plots)))
    settings = ip._prepare_settings({"number_of_plots": 1})
    ip._generate_plots(np.zeros((3, 1, 1)), 7, [np.zeros((1, 1), bool), None, np.zeros((1, 1), bool)], [0, np.nan, 0],
                       [0.0, 9.0, 0.0], settings, show_plots=False, save_plots=True, output_folder="o")
    assert called == [("vis", 0, 7, "o"), ("counts", 7, False), ("heat", 7, True)]
    called.clear()
    ip._generate_plots(np.zeros((3, 1, 1)), 7, [None] * 3, [0] * 3, [0.0] * 3, settings, False, False, "o",
                       False, True, False)                                   # everything positional
    assert called == [("counts", 7, False)]


def test_detect_pipeline_with_every_stage_stubbed(ip, monkeypatch):            # t.py:496-556
    frames = [np.zeros((1, 1)), np.ones((1, 1))]

    class Settings:
        flow_threshold = 5.0
        sliding_window_radius = 1
        brightness_threshold = 128
        static_threshold = 50
        consecutive_threshold = 2
        max_cluster_size = 10
        number_of_plots = 1

    monkeypatch.setattr(ip, "_prepare_settings", lambda params: Settings)
    monkeypatch.setattr(ip, "compute_optical_flow_metric", lambda fs: [0.0, 10.0])
    monkeypatch.setattr(ip, "compute_background", lambda frames, radius: np.array([np.zeros((1, 1)), np.zeros((1, 1))]))
    monkeypatch.setattr(ip, "_raw_damaged_mask", lambda frames, brightness: np.array([[True]]))
    monkeypatch.setattr(ip, "remove_bright_regions",
                        lambda background, bright_threshold, raw_mask, max_cluster_size: np.array([[False]]))
    monkeypatch.setattr(ip, "compute_persistent_mask", lambda masks, consecutive_threshold: np.array([[False]]))
    monkeypatch.setattr(ip, "filter_consecutive_pixels", lambda masks, persistent: ([np.array([[False]]), None], [0, np.nan]))
    monkeypatch.setattr(ip, "compute_static_mask",
                        lambda frames, background, brightness_threshold, static_threshold, min_valid_frames: np.array([[False]]))
    monkeypatch.setattr(ip, "apply_static_suppression",
                        lambda masks, persistent, static_mask: ([np.array([[False]]), None], np.array([0.0, np.nan])))
    monkeypatch.setattr(ip, "find_bright_area_estimates", lambda frames, mask, brightness_threshold: [1.0, np.nan])
    monkeypatch.setattr(ip, "_get_final_count", lambda frames, bright_estimates: np.array([1.0, np.nan]))
    monkeypatch.setattr(ip, "compute_cluster_stats",
                        lambda frames, masks, flows, settings: (np.array([0.0, 0.0]), np.array([0.0, 0.0]), np.array([0.0, 0.0])))
    called = {}
    monkeypatch.setattr(ip, "_generate_plots", lambda *args, **kwargs: called.setdefault("plot", True))
    results = ip.detect_damaged_pixels(frames, plot=False)
    assert isinstance(results, tuple) and len(results) == 4 and "plot" not in called
    total, cc, cs, cb = ip.detect_damaged_pixels(frames, plot=True)
    assert "plot" in called
    assert np.allclose(total, [1.0, np.nan], equal_nan=True) and np.array_equal(cc, np.array([0.0, 0.0]))


# ---- command line (ip.py:1250-1316) ---------------------------------------------------------------------------
def test_cli_defaults_match_the_reference(ip):
    o = ip._cli_options([])
    kw = o["main_kwargs"]
    assert o["video_filename"] == "11_01_H_170726081325.avi"                  # ip.py:1289
    assert kw["max_chunks"] == 2 and kw["step_size"] == 1000 and kw["average_time"] == 1.0   # ip.py:1291-1294
    assert kw["plot"] is False and kw["show_plots"] is False and kw["save_plots"] is False
    assert kw["visualise_damaged_pixels_plot"] is False                       # ip.py:1300-1301: False on the CLI
    assert kw["plot_damaged_pixels_plot"] is True and kw["plot_heatmap_plot"] is True
    assert kw["output_folder"] == "results" and kw["checkpoint_folder"] is None and o["average_plots"] is True
    assert kw["params"] == dict(consecutive_threshold=2, brightness_threshold=170, flow_threshold=2.0, static_threshold=50,
                                min_cluster_size=5, max_cluster_size=20, min_circularity=0.1, sliding_window_radius=3,
                                number_of_plots=20)


def test_cli_string_flags(ip):
    # what image_processing_script.sh:116 forwards: every value a string
    o = ip._cli_options(["--video_filename", "v.avi", "--average_time", "0.5", "--max_chunks", "none", "--step_size", "250",
                         "--show_plots", "False", "--save_plots", "TRUE", "--chunk_plots", "true", "--average_plots", "false",
                         "--visualise_damaged_pixels_plot", "True", "--plot_heatmap_plot", "false", "--output_folder", "out",
                         "-ct", "3", "-bt", "180", "-ft", "1.5", "-st", "40", "-mncs", "4", "-mxcs", "25", "-mnc", "0.2",
                         "-swr", "2", "-np", "5"])
    kw = o["main_kwargs"]
    assert o["video_filename"] == "v.avi" and kw["max_chunks"] is None and kw["step_size"] == 250 and kw["average_time"] == 0.5
    assert kw["save_plots"] is True and kw["show_plots"] is False and kw["plot"] is True and o["average_plots"] is False
    assert kw["visualise_damaged_pixels_plot"] is True and kw["plot_heatmap_plot"] is False and kw["output_folder"] == "out"
    assert kw["params"] == dict(consecutive_threshold=3, brightness_threshold=180, flow_threshold=1.5, static_threshold=40,
                                min_cluster_size=4, max_cluster_size=25, min_circularity=0.2, sliding_window_radius=2,
                                number_of_plots=5)
    assert ip._cli_options(["-mc", "7"])["main_kwargs"]["max_chunks"] == 7
    # plots per chunk only when something is shown or saved (ip.py:1304)
    assert ip._cli_options(["--chunk_plots", "True"])["main_kwargs"]["plot"] is False


def test_cli_runs_main_and_prints(ip, monkeypatch, capsys):
    got = {}

    def fake_main(video_filename, **kw):
        got.update(kw, video_filename=video_filename)
        return dict(averages_counts=[1.0], averages_clusters=[2.0], averages_size=[3.0], averages_brightness=[4.0],
                    times=np.array([0.5]))
    monkeypatch.setattr(ip, "main", fake_main)
    res = ip._cli(["-vf", "x.avi", "--average_plots", "False"])
    out = capsys.readouterr().out
    assert got["video_filename"] == "x.avi" and got["max_chunks"] == 2 and res["averages_size"] == [3.0]
    assert "counts: [1.0]" in out and "Done" in out


# ---- plot functions: Agg smoke, output file names of the reference ------------------------------------------------
def test_plot_functions_write_the_reference_file_names(ip, tmp_path):
    if not ip.HAS_MATPLOTLIB:
        pytest.skip("matplotlib not importable")
    out = str(tmp_path / "plots")
    frame = (np.arange(48 * 64).reshape(48, 64) % 251).astype(np.uint8)
    mask = np.zeros((48, 64), bool)
    mask[5:7, 9:12] = True
    ip.visualise_damaged_pixels(frame, 3, 1, mask, 6, show_plots=False, save_plots=True, output_folder=out)
    ip.visualise_damaged_pixels(frame, 4, 1, None, float("nan"), show_plots=False, save_plots=True, output_folder=out)
    ip.plot_heatmap(np.random.default_rng(0).random((48, 64)) * 100, 1, show_plots=False, save_plots=True, output_folder=out)
    ip.plot_damaged_pixels([3, 4, float("nan"), 6], 1, show_plots=False, save_plots=True, output_folder=out)
    ip.plot_all_chunks([1, 2, 3], [0, 1, 0], [0, 5.5, 0], [np.nan, 200.0, np.nan], np.array([0.5, 1.5, 2.5]),
                       show_plots=False, save_plots=True, output_folder=out)
    ip.plot_all_chunks([1, 2], [0, 1], [0, 5.5], [np.nan, 200.0], None, show_plots=False, save_plots=False, output_folder=out)
    names = sorted(os.listdir(out))
    assert names == sorted(["visualise_damaged_pixels_frame_chunk_index=1_frame_index=3.png",       # ip.py:999
                            "visualise_damaged_pixels_frame_chunk_index=1_frame_index=4.png",
                            "damaged_pixel_heatmap_chunk_index=1.png",                                # ip.py:1029
                            "plot_damged_pixels_chunk_index=1.png",                                   # ip.py:1062 (sic)
                            "combined_chunks_overview.png"])                                          # ip.py:1127
    assert all(os.path.getsize(os.path.join(out, n)) > 1000 for n in names)
    # newer generation: no chunk index (t.py:443-451)
    out2 = str(tmp_path / "plots2")
    ip.visualise_damaged_pixels(frame, 0, mask, 6, show_plots=False, save_plots=True, output_folder=out2)
    ip.plot_damaged_pixels([1, 2], show_plots=False, save_plots=True, output_folder=out2)
    ip.plot_heatmap(np.zeros((4, 4)), show_plots=False, save_plots=True, output_folder=out2)
    assert len(os.listdir(out2)) == 3


def test_blend_matches_addweighted_rounding(ip):
    base = np.array([[0, 1, 100, 255]], np.uint8)
    over = np.array([[0, 255, 165, 255]], np.uint8)
    # cv2.addWeighted(base, 0.8, over, 1.0, 0): saturate_cast<uchar>(round(0.8 * base + over))
    assert ip._blend(base, over).tolist() == [[0, 255, 245, 255]]


# ---- checkpoint fingerprint ---------------------------------------------------------------------------------------
def test_checkpoint_fingerprint_depends_on_video_chunking_and_settings(ip, tmp_path):
    a = tmp_path / "a.avi"
    a.write_bytes(b"x" * 100)
    b = tmp_path / "b.avi"
    b.write_bytes(b"x" * 101)
    fp = ip._checkpoint_fingerprint
    base = fp(str(a), 300, 100, None)
    assert base == fp(str(a), 300, 100, {}) == fp(str(a), 300, 100, {"flow_threshold": 2.0, "number_of_plots": 3})
    assert base != fp(str(b), 300, 100, None)                                    # another file
    assert base != fp(str(a), 301, 100, None) and base != fp(str(a), 300, 50, None)
    assert base != fp(str(a), 300, 100, {"brightness_threshold": 171})
    os.utime(str(a), ns=(1, 1))
    assert base != fp(str(a), 300, 100, None)                                    # same name and size, rewritten

"""Multi-video batch: V videos over the GPUs of one node, one worker process per GPU, no collectives.

BASELINE.json configs[4] ("4K multi-video batch, 8 videos concurrently across 8 x MI355X with overlapped
decode / H2D copies").  The reference's unit of independence is the chunk, and so the video
(ip.py:1204-1227: every chunk is loaded and processed on its own; nothing is carried from one to the next):
different videos therefore go to different GPUs with no exchange at all, and each GPU runs the single-GPU
chunk driver ``image_processing.main`` unchanged -- a decode thread filling pinned memory for chunk k+1, the
upload on a side stream, ``ipx_detect_chunk`` enqueued before chunk k is post-processed.  Only the result
dictionaries (a few floats per second of video) travel back to the launcher.

    python -m image_processing_amd.multi --videos a.avi,b.avi,... --gpus N [flags of image_processing.py]

Process model.  Workers are FRESH processes that initialise the GPU themselves; none is created by replacing
the program of a process that already holds GPU state.  They are forked from a fork server (multiprocessing's
"forkserver" start method): a small helper process that never touches the GPU and that must itself be
started before the launching process initialises the GPU (``ensure_launcher``; the command line does it first
thing, a test suite does it at configure time).  Worker w owns device ``devices[w]`` and processes videos
w, w + W, w + 2W, ... in order, so a GPU never runs two of this launcher's videos at once unless a device is
listed twice (which is how the path is rehearsed on a one-GPU box).
"""
import multiprocessing as mp
import multiprocessing.connection
import os
import sys
import time
import traceback

_STARTED = False


def _gpu_initialised():
    t = sys.modules.get("torch")
    try:
        return bool(t is not None and t.cuda.is_initialized())
    except Exception:                                   # noqa: BLE001
        return False


def ensure_launcher():
    """Start the fork server the workers are forked from (idempotent).  Call it before this process touches
    the GPU: afterwards a new helper could only be made by fork + exec from a process with GPU state, which
    this module refuses to do."""
    global _STARTED
    if _STARTED:
        return
    if _gpu_initialised():
        raise RuntimeError("image_processing_amd.multi.ensure_launcher() must run before this process initialises "
                           "the GPU (call it at start-up)")
    from multiprocessing import forkserver
    forkserver.ensure_running()
    _STARTED = True


def run_videos(videos, gpus=None, devices=None, timeout=None, per_video_output=True, **main_kwargs):
    """Process ``videos`` (file names) with ``image_processing.main(video, **main_kwargs)``, one worker process
    per entry of ``devices`` (default: device ordinals 0..gpus-1, ``gpus`` defaulting to the number of GPUs).
    With ``per_video_output`` (default) ``output_folder`` / ``checkpoint_folder`` become one sub-folder per video.
    Returns a list, in the order of ``videos``, of dicts: main's result (plain lists) plus ``video``,
    ``device`` and ``seconds``.  Raises RuntimeError if a worker fails or ``timeout`` seconds pass."""
    videos = list(videos)
    if not videos:
        return []
    if devices is None:
        if gpus is None:
            import torch
            gpus = torch.cuda.device_count()            # counting devices does not initialise the GPU
        devices = list(range(int(gpus)))
    devices = [int(d) for d in devices][:len(videos)]
    if not devices:
        raise RuntimeError("no GPU to run on")
    ensure_launcher()
    ctx = mp.get_context("forkserver")
    workers = []
    for w, dev in enumerate(devices):
        jobs = [(i, videos[i]) for i in range(w, len(videos), len(devices))]
        parent, child = ctx.Pipe(duplex=False)
        p = ctx.Process(target=_worker, args=(w, dev, jobs, dict(main_kwargs), bool(per_video_output), child), daemon=True)
        p.start()
        child.close()
        workers.append((p, parent))
    results = [None] * len(videos)
    deadline = None if timeout is None else time.monotonic() + timeout
    failure = None
    live = {id(c): (p, c) for p, c in workers}
    try:
        while live and failure is None:
            left = None if deadline is None else max(0.0, deadline - time.monotonic())
            ready = mp.connection.wait([c for _, c in live.values()], timeout=left)
            if not ready:
                failure = "timed out after %s s" % timeout
                break
            for c in ready:
                try:
                    kind, a, b, secs, dev = c.recv()
                except EOFError:
                    p = live.pop(id(c))[0]
                    p.join(5)
                    if p.exitcode not in (0, None):
                        failure = "a worker died with exit code %r" % p.exitcode
                    continue
                if kind == "ok":
                    b.update(video=videos[a], device=dev, seconds=secs)
                    results[a] = b
                elif kind == "error":
                    failure = "worker %d failed:\n%s" % (a, b)
                elif kind == "done":
                    live.pop(id(c))[0].join(30)
    finally:
        for p, c in workers:
            if p.is_alive():
                p.terminate()                           # exactly the processes this call started
                p.join(10)
            c.close()
    if failure is None and any(r is None for r in results):
        failure = "no result for video(s) %s" % [videos[i] for i, r in enumerate(results) if r is None]
    if failure:
        raise RuntimeError("multi-video batch: " + failure)
    return results


def video_output_folder(base, video):
    """<base>/<file name of the video without its extension>: plots of different videos do not overwrite each other."""
    return os.path.join(base, os.path.splitext(os.path.basename(video))[0])


def _worker(slot, device, jobs, main_kwargs, per_video_output, conn):
    """One GPU's share of the batch, in a fresh process: ``jobs`` = [(index, video filename), ...].  With
    ``per_video_output`` every video gets its own output (and checkpoint) folder below the one given."""
    try:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("worker %d: no GPU visible" % slot)
        torch.cuda.set_device(int(device))
        import image_processing as ip
        for j, (index, video) in enumerate(jobs):
            # the next video's first chunk is decoded and uploaded under this one's last chunk (image_processing.queue_next_video)
            ip.queue_next_video(jobs[j + 1][1] if j + 1 < len(jobs) else None, main_kwargs.get("step_size") or 1000)
            kw = dict(main_kwargs)
            for key in ("output_folder", "checkpoint_folder"):
                if per_video_output and kw.get(key):
                    kw[key] = video_output_folder(kw[key], video)
            t0 = time.perf_counter()
            res = ip.main(video, **kw)
            torch.cuda.synchronize()
            res = {k: (v.tolist() if hasattr(v, "tolist") else (list(v) if isinstance(v, (list, tuple)) else v))
                   for k, v in res.items()}
            conn.send(("ok", index, res, time.perf_counter() - t0, int(device)))
        conn.send(("done", slot, None, 0.0, int(device)))
    except BaseException:                               # noqa: BLE001
        try:
            conn.send(("error", slot, traceback.format_exc(), 0.0, int(device)))
        except Exception:                               # noqa: BLE001
            pass
    finally:
        conn.close()


def _cli(argv=None):
    import argparse
    ensure_launcher()                                   # before anything can touch the GPU
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0], add_help=False)
    ap.add_argument("--videos", required=True, help="comma-separated video files")
    ap.add_argument("--gpus", type=int, default=None, help="worker processes = GPUs to use (default: all)")
    ap.add_argument("--devices", default=None, help="comma-separated device ordinals, one worker each (overrides --gpus)")
    ap.add_argument("--timeout", type=float, default=None)
    args, rest = ap.parse_known_args(argv)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import image_processing as ip
    opts = ip._cli_options(rest)                        # the reference's flags (ip.py:1250-1316), video_filename unused
    opts["main_kwargs"].pop("gpus", None)               # here --gpus counts worker processes; every worker drives one device
    videos = [v for v in args.videos.split(",") if v]
    devices = [int(d) for d in args.devices.split(",")] if args.devices else None
    t0 = time.perf_counter()
    results = run_videos(videos, gpus=args.gpus, devices=devices, timeout=args.timeout, **opts["main_kwargs"])
    dt = time.perf_counter() - t0
    for r in results:
        print("%s (GPU %d, %.1f s)" % (r["video"], r["device"], r["seconds"]))
        if opts["average_plots"] and ip.HAS_MATPLOTLIB and (opts["show_plots"] or opts["save_plots"]):
            ip.plot_all_chunks(r["averages_counts"], r["averages_clusters"], r["averages_size"], r["averages_brightness"],
                               r["times"], show_plots=opts["show_plots"], save_plots=opts["save_plots"],
                               output_folder=video_output_folder(opts["main_kwargs"]["output_folder"], r["video"]))
        else:
            for key in ("averages_counts", "averages_clusters", "averages_size", "averages_brightness", "times"):
                print("  %s: %s" % (key, r[key]))
    print("\nDone: %d videos in %.1f s" % (len(results), dt))
    return results


if __name__ == "__main__":
    _cli()

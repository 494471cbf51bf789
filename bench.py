#!/usr/bin/env python3
"""bench.py -- frames/s of the damaged-pixel detection path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the whole path (ipx_detect_chunk's stages, ip.py:526-632) over one chunk of
synthetic 1920x1080 uint8 frames already resident in HBM, results copied back and turned into the
reference's four float64 arrays.  N = 1: BASELINE.json configs[1] ("1 MI355X, 1920x1080 synthetic
video", processed in the reference's 1000-frame chunks).  N > 1 (one rank per GPU): configs[2], a chunk of
N x 1000 frames is sharded by frame range, 1000 frames per GPU (weak scaling), with the persistence /
static-map / result exchanges over RCCL inside the step.

Launching N > 1.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`:
WORLD_SIZE is set) this process IS a rank.  Started plainly (`python bench.py --gpus N`, no WORLD_SIZE) it starts
that launcher itself as a fresh child process -- before this process has imported torch or touched the GPU --
relays rank 0's JSON line and exits with the child's status.

Rank 0 prints ONE JSON line.
* `value`: K steps timed with the built-in profiler OFF.
* `roofline`, `kernels`: a SECOND pass of the same steps with libipx's profiler on (HIP events on the launch
  stream around every kernel); the dominant kernel class (most GPU time) is priced as algorithmic bytes / time
  against the 8 TB/s HBM peak (`frac`), and as the HBM bytes of the committed PMC passes / time (`frac_hbm`).
* `cpu_baseline`: the CPU oracle (oracle/, a port: the reference needs cv2 + numba, absent from this image)
  on a bounded sample of the same frames, rank 0, N = 1 only, with per-stage milliseconds.
* `config.min_flow_margin`: how close the motion test came to its threshold (the discard list's robustness).
* `throughput_pinned_host`, `throughput_from_avi` (N = 1, never `value`): the same path with chunks starting in
  pinned host memory (upload on a side stream under the previous chunk) and in an uncompressed AVI on disk
  (decode thread + the above), SURVEY section 8(d) figures (ii) and (iii).  --no-modes skips them.
* N > 1: `collectives`: GPU milliseconds, calls and bytes of the RCCL exchanges per chunk; `strong_1000`: the
  reference's own 1000-frame chunk (ip.py:1186-1192) split N ways, 1000 / N frames per GPU (strong scaling;
  what a drop-in user's single video gets from N GPUs through intra-chunk sharding), same timing rules.

Other workloads (never the default): --height/--width/--frames, --motion (configs[3]), and
--videos V (configs[4]): V synthetic videos of --video-frames frames in uncompressed AVI files, every rank (= GPU
process) runs the chunk driver image_processing.main on its share (videos rank, rank + N, ...) with overlapped
decode / upload, no collectives; a step is one pass over all V videos.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2,
                    help="untimed steps first (default 2: the loop is pipelined over two chunks and two result slots, and the first step of each still sets buffers up: tools/step_times.py)")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default 1000)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--motion", action="store_true", help="camera-motion-heavy sequence (configs[3])")
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled pass (roofline = null)")
    ap.add_argument("--profile-steps", type=int, default=2, help="steps of the profiled pass")
    ap.add_argument("--no-modes", action="store_true", help="skip the pinned-host and from-AVI throughput figures")
    ap.add_argument("--strong-frames", type=int, default=1000,
                    help="N > 1: frames of the chunk that is also timed split N ways (strong scaling; 0 = skip)")
    ap.add_argument("--group", type=int, default=0,
                    help="one process, N ranks inside libipx (ipx_group_submit / ipx_group_wait): chunks start in pinned host memory")
    ap.add_argument("--group-devices", default=None, help="comma-separated device ordinals of the ranks (default: 0..N-1, or N times device 0 on a smaller node)")
    ap.add_argument("--group-transport", default="auto", choices=["auto", "peer", "rccl"])
    ap.add_argument("--videos", type=int, default=0, help="configs[4]: number of videos of the multi-video batch")
    ap.add_argument("--video-frames", type=int, default=240, help="frames per video (--videos)")
    ap.add_argument("--video-chunk", type=int, default=120, help="step_size of the chunk driver (--videos)")
    return ap.parse_args()


def load_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*_traffic.json), or None.
    The newest file (by name) that names this workload wins."""
    import glob
    best = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            with open(p) as f:
                d = json.load(f)
        except Exception:
            continue
        if d.get("workload") == workload and kernel in d.get("kernels", {}):
            best = d["kernels"][kernel].get("hbm_bytes_per_launch")
    return best


def cpu_threads():
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    return min(omp, avail) if omp > 0 else avail


def cpu_baseline(frames_np):
    """The oracle pipeline on the sample, timed on this host's cores (OpenMP inside the C part)."""
    from oracle import pipeline
    stages = {}
    t0 = time.perf_counter()
    pipeline.detect_damaged_pixels(frames_np, timings=stages)
    dt = time.perf_counter() - t0
    threads = cpu_threads()
    n, h, w = frames_np.shape
    return dict(value=n / dt, unit="frames/s", cores=threads, threads=threads, kind="port",
                sample="%d frames of the step's first chunk (%dx%d), whole path incl. flow; %.1f s" % (n, w, h, dt),
                per_stage_ms={k: round(1e3 * v, 1) for k, v in stages.items()},
                note="the reference's batch script asks for 36 cores (image_processing_script.sh:5); this box gives the "
                     "job %d, so there is no 36-thread line" % threads)


def device_copy_gbs(src):
    """Measured device-copy ceiling beside the 8 TB/s spec: numel bytes read + numel written per copy."""
    import torch
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2.0 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9


def throughput_modes(chunks_dev, H, W):
    """SURVEY 8(d) figures (ii) and (iii) for the chunks of the main run; returns two dicts."""
    import numpy as np
    import torch
    import image_processing as ip
    from image_processing_amd import avi
    n = chunks_dev[0].shape[0]
    pinned = []
    for b in chunks_dev:
        p = torch.empty(b.shape, dtype=torch.uint8, pin_memory=True)
        p.copy_(b)
        pinned.append(p)
    torch.cuda.synchronize()
    kw = dict(plot=False, show_plots=False, save_plots=False, output_folder="results", visualise_damaged_pixels_plot=True,
              plot_damaged_pixels_plot=True, plot_heatmap_plot=True, params=None)

    delivered = []

    def run_pinned(count):
        pending = None
        for i in range(count):
            done = ip._submit_chunk(([None] * n, pinned[i % 2]), i, kw)
            if pending is not None:
                pending()
                delivered.append(time.perf_counter())
            pending = done
        pending()
        delivered.append(time.perf_counter())
    run_pinned(1)
    torch.cuda.synchronize()
    count = 4
    del delivered[:]
    t0 = time.perf_counter()
    run_pinned(count)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from_pinned = dict(value=round(count * n / dt, 1), unit="frames/s", chunks=count, frames_per_chunk=n,
                       host_to_hbm_gbs=round(count * n * H * W / dt / 1e9, 2),
                       steady_state=round((count - 1) * n / (delivered[-1] - delivered[0]), 1),
                       what="chunks start in pinned host memory; upload on a side stream under the previous chunk.  value = the "
                            "whole run, whose first upload nothing overlaps; steady_state = delivery to delivery from the second chunk on")
    # (iii): an uncompressed 8-bit AVI on disk (a RIFF AVI holds 4 GB), read by image_processing.main
    nch = 4
    per = max(2, min(n, int(3.8e9 / (H * W)) // nch))
    host = [p.numpy() for p in pinned]
    fd, fn = tempfile.mkstemp(suffix=".avi", prefix="ipx_bench_")
    os.close(fd)
    try:
        avi.write_avi(fn, [host[c % 2][i] for c in range(nch) for i in range(per)], fps=25.0)
        import contextlib
        import io
        def timed_pass():
            ip._TIMELINE = tl = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = ip.main(fn, average_time=1.0, max_chunks=None, step_size=per)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ip._TIMELINE = None
            assert len(res["averages_counts"]) > 0 and np.isfinite(np.nansum(res["averages_counts"]))
            tl.sort()
            # steady state: from the delivery of the first chunk to the delivery of the last (what every further chunk of
            # a longer video costs; the first chunk's decode, which nothing overlaps, and the start-up are left out)
            steady = (len(tl) - 1) * per / (tl[-1][2] - tl[0][2]) if len(tl) > 1 else None
            return nch * per / dt, steady
        with contextlib.redirect_stdout(io.StringIO()):
            first, first_steady = timed_pass()           # fresh staging buffers are pinned during this one
            again, steady = timed_pass()
        from_avi = dict(value=round(again, 1), unit="frames/s", chunks=nch, frames_per_chunk=per,
                        first_pass=round(first, 1), steady_state=round(steady, 1) if steady else None,
                        first_pass_steady_state=round(first_steady, 1) if first_steady else None,
                        what="chunks start in an uncompressed 8-bit AVI (page cache): large reads into a ring of pinned staging "
                             "buffers, upload on a side stream, rows turned over on the GPU (ipx_unpack_dib).  value = second pass "
                             "over the file, whole run incl. the first chunk's decode, which nothing overlaps; first_pass = the "
                             "first run in the process (pins the staging ring); steady_state = chunks 2.. of the second pass "
                             "(delivery to delivery: what each further chunk of a longer video costs)")
    finally:
        os.remove(fn)
    return from_pinned, from_avi


def bench_videos(args, world, rank, dev):
    """configs[4]: the multi-video batch.  Every rank owns videos rank, rank + N, ...; no collectives on the data path."""
    import contextlib
    import io
    import numpy as np
    import torch
    import torch.distributed as dist
    import image_processing as ip
    from image_processing_amd import avi, synth
    H, W, M, S = args.height, args.width, args.video_frames, args.video_chunk
    mine = list(range(rank, args.videos, world))
    tmp = tempfile.mkdtemp(prefix="ipx_videos_")
    files = []
    try:
        for v in mine:                                     # synthetic footage -> uncompressed AVI (the decode side's input)
            fr = synth.chunk_torch(4 + v, M, H, W, dev).cpu().numpy()
            fn = os.path.join(tmp, "video%02d.avi" % v)
            avi.write_avi(fn, fr, fps=25.0)
            files.append(fn)
            del fr

        def one_pass():
            out = []
            with contextlib.redirect_stdout(io.StringIO()):
                for k, fn in enumerate(files):             # (as image_processing_amd.multi's workers do)
                    ip.queue_next_video(files[k + 1] if k + 1 < len(files) else None, S)
                    out.append(ip.main(fn, average_time=1.0, max_chunks=None, step_size=S))
            return out

        def fence():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        for _ in range(args.warmup):
            one_pass()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = one_pass()
        fence()
        dt = time.perf_counter() - t0
    finally:
        for fn in files:
            if os.path.exists(fn):
                os.remove(fn)
        os.rmdir(tmp)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        total = args.videos * M * args.steps
        ok = all(np.isfinite(np.nansum(r["averages_counts"])) for r in res)
        line = dict(metric="frames/sec (whole node), multi-video batch", value=round(total / dt, 2), unit="frames/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 3),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8 (flow metric in f32/f64)",
                    data="synthetic",
                    config=dict(workload="%d videos of %dx%d u8, %d frames each in uncompressed AVI files, %d-frame chunks, "
                                         "one process per GPU, decode + upload overlapped, no collectives" % (args.videos, W, H, M, S),
                                videos=args.videos, videos_per_gpu=len(mine), results_finite=bool(ok)),
                    roofline=None, cpu_baseline=None,
                    note="includes decode from the page cache and the PCIe upload: not roofline-comparable")
        print(json.dumps(line))


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: run the N ranks under torch.distributed.run in a child process
    (this process has not imported torch, so nothing here holds GPU state), relay the JSON line, return its status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith("{") and t.endswith("}"):
            line = t                                   # rank 0's result (the last such line wins)
        else:
            sys.stderr.write(ln)                       # anything else a rank printed
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


def bench_group(args):
    """--group N: the in-library multi-GPU path as a single-process caller (a C program, the reference's own module) sees it:
    one chunk of --frames frames in PINNED HOST memory per step, cut by frame range over N ranks inside libipx, two chunks
    in flight (chunk k+1 uploads under chunk k).  The figure includes the PCIe upload: never `value` of the default run."""
    import numpy as np
    import torch
    from image_processing_amd import hostmath, synth
    from image_processing_amd.engine import Group
    N = args.group
    have = torch.cuda.device_count()
    if args.group_devices:
        devices = [int(x) for x in args.group_devices.split(",")]
    else:
        devices = list(range(N)) if have >= N else [0] * N
    H, W, n = args.height, args.width, args.frames or 1000
    dev0 = torch.device("cuda", devices[0])
    speeds = [0, 0.5, 4, 8] if args.motion else None
    pinned = []
    for c in range(2):
        fr = synth.chunk_torch(1000 * c + 2, n, H, W, dev0, speeds=speeds)
        p = torch.empty(fr.shape, dtype=torch.uint8, pin_memory=True)
        p.copy_(fr)
        pinned.append(p)
        del fr
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    g = Group(devices, transport=args.group_transport)

    delivered = []

    def run(first, count):
        res, pend = [], None
        for i in range(first, first + count):
            h = g.submit(pinned[i % 2].numpy())        # the upload starts at once, under the chunk before
            if pend is not None:
                res.append(g.wait(pend))
                delivered.append(time.perf_counter())
            pend = h
        if pend is not None:
            res.append(g.wait(pend))
            delivered.append(time.perf_counter())
        return res

    run(0, args.warmup)
    del delivered[:]
    t0 = time.perf_counter()
    results = run(args.warmup, args.steps)
    dt = time.perf_counter() - t0
    # (the timed steps start with an empty pipeline: the first chunk's upload is not overlapped.  Delivery to delivery
    # is what every further chunk costs)
    steady = (len(delivered) - 1) * n / (delivered[-1] - delivered[0]) if len(delivered) > 1 else None
    thr = float(g.params.flow_threshold)
    margins = [hostmath.flow_margin(r["flows"], thr) for r in results]
    line = dict(metric="frames/sec (whole node) at 1080p, from pinned host memory through ipx_group_*",
                value=round(n * args.steps / dt, 2), unit="frames/s", n_gpus=len(set(devices)), ranks=N, steps=args.steps,
                warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 3), higher_is_better=True, scaling="strong",
                vs_baseline=None, dtype="u8 (flow metric in f32/f64)", data="synthetic",
                config=dict(workload="%dx%d u8, %d-frame chunks in pinned host memory, cut by frame range over %d ranks inside "
                                     "libipx (devices %s, transport %s), two chunks in flight" % (W, H, n, N, devices, g.transport),
                            frames_per_rank=n // N, host_to_hbm_gbs=round(n * H * W * args.steps / dt / 1e9, 2),
                            steady_state=round(steady, 1) if steady else None,
                            deliveries_ms=[round(1e3 * (t - t0), 1) for t in delivered],
                            kept_frames_last_step=int(np.sum(results[-1]["keep"])),
                            min_flow_margin=round(min(m["min_margin"] for m in margins), 6)),
                roofline=None, cpu_baseline=None,
                note="includes the PCIe upload of every frame: not roofline-comparable; the resident figure is the default run's `value`")
    g.close()
    print(json.dumps(line))


def main():
    args = parse()
    if args.group > 0:
        return bench_group(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import numpy as np
    import torch
    import torch.distributed as dist
    from image_processing_amd import hostmath, sharding, synth
    from image_processing_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box: IPX_BENCH_REHEARSE=1 puts every rank on GPU 0 over gloo
    rehearse = os.environ.get("IPX_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if args.videos > 0:
        bench_videos(args, world, rank, dev)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    H, W = args.height, args.width
    F = args.frames or 1000                  # the same work per GPU at every N (weak scaling)
    n_total = F * world
    eng = Engine(local_rank)
    comm = sharding.make_comm()
    radius = int(eng.params.sliding_window_radius)
    speeds = [0, 0.5, 4, 8] if args.motion else None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_job(n_frames):
        """Two distinct synthetic chunks of `n_frames` frames in HBM, this rank's frame range + halo of each, and
        submit(i): enqueue step i, return a callable that completes it (copy-back + the host's float glue)."""
        a, b = sharding.shard_range(n_frames, rank, world)
        lo, hi = sharding.halo_range(n_frames, a, b, radius)
        chunks = []
        for c in range(2):
            own = synth.chunk_torch(1000 * c + 2 + rank, b - a, H, W, dev, speeds=speeds)
            if world > 1:   # halo frames come from the neighbours' ranges (stands in for decoding them)
                r = max(radius, 1)
                edges = comm.all_gather(torch.cat([own[:r], own[-r:]]).contiguous())
                parts = []
                if rank > 0:
                    parts.append(edges[rank - 1][r:][-(a - lo):])
                parts.append(own)
                if rank + 1 < world:
                    parts.append(edges[rank + 1][:r][:hi - b])
                own = torch.cat(parts).contiguous()
            chunks.append(own)

        def submit(i):
            fr = chunks[i % 2]
            if world == 1:
                return eng.detect_chunk_async(fr).result
            out = sharding.detect_chunk_sharded(eng, fr, a, b, n_frames, comm)
            return lambda: sharding.results_to_host(out, H * W)
        return chunks, submit

    def run(submit, first, count):
        """`count` steps, software-pipelined as a chunk driver would: step k+1 is on the GPU while the host
        finishes step k.  Every step's results are produced inside the call."""
        res, pending = [], None
        for i in range(first, first + count):
            nxt = submit(i)
            if pending is not None:
                res.append(pending())
            pending = nxt
        if pending is not None:
            res.append(pending())
        return res

    def timed(submit):
        """W untimed steps, then K steps between fences; the slowest rank's time."""
        run(submit, 0, args.warmup)
        fence()
        t0 = time.perf_counter()
        res = run(submit, args.warmup, args.steps)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return res, dt

    # ---- pass 1: the timed steps, profiler off ------------------------------------------------------
    chunks, submit = make_job(n_total)
    results, dt = timed(submit)

    # ---- pass 2: the same steps with per-kernel events (roofline, kernel table, collective times) -----
    prof, prof_levels, coll, psteps, dt_prof = {}, {}, None, max(1, args.profile_steps), None
    if not args.no_profile:
        eng.profile_enable(True)
        eng.profile_reset()
        if world > 1 and hasattr(comm, "timed"):
            comm.collective_ms()
            comm.timed = True
        fence()
        t1 = time.perf_counter()
        run(submit, args.warmup, psteps)
        fence()
        dt_prof = time.perf_counter() - t1
        prof = eng.profile_read()
        prof_levels = eng.profile_read_flow_levels()
        eng.profile_enable(False)
        if world > 1 and hasattr(comm, "timed"):
            calls, nbytes = comm.calls, comm.bytes
            ms = comm.collective_ms()
            comm.timed = False
            coll = dict(ms_per_chunk=round(ms / psteps, 3), calls_per_chunk=calls // psteps, bytes_per_chunk=nbytes // psteps,
                        backend="gloo (rehearsal)" if rehearse else "nccl (RCCL)", rank=0)

    # ---- N > 1: the reference's 1000-frame chunk split N ways (strong scaling) ------------------------
    strong = None
    S = args.strong_frames
    if world > 1 and S > 0 and S // world >= max(2, int(eng.params.consecutive_threshold)):
        chunks_s, submit_s = make_job(S)
        res_s, dt_s = timed(submit_s)
        strong = dict(value=round(S * args.steps / dt_s, 2), unit="frames/s", ms_per_step=round(1e3 * dt_s / args.steps, 3),
                      scaling="strong", frames_per_chunk=S, frames_per_gpu=S // world,
                      what="one %d-frame chunk (the reference's step_size, ip.py:1186-1192) sharded by frame range "
                           "over %d GPUs" % (S, world))
        del chunks_s, submit_s, res_s

    if rank == 0:
        copy_gbs = device_copy_gbs(chunks[0])
        workload = "%dx%d u8, %d-frame chunks%s" % (W, H, n_total, ", camera motion" if args.motion else "")
        roof = None
        kernels = {}
        if prof:
            tot_ms = sum(v["ms"] for v in prof.values())
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                kernels[k] = dict(launches=v["launches"], ms_total=round(v["ms"], 3),
                                  avg_us=round(1e3 * v["ms"] / v["launches"], 2),
                                  gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None,
                                  share=round(v["ms"] / tot_ms, 4))
            dom = max(prof, key=lambda k: prof[k]["ms"])
            v = prof[dom]
            avg_s = v["ms"] * 1e-3 / v["launches"]
            ach = v["bytes"] / v["launches"] / avg_s / 1e9
            traffic = load_traffic(dom, workload)
            roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                        achieved_hbm=round(traffic / avg_s / 1e9, 1) if traffic else None,
                        frac_hbm=round(traffic / avg_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                        bytes_per_launch=round(v["bytes"] / v["launches"]), avg_launch_us=round(1e6 * avg_s, 2),
                        launches=v["launches"], share_of_gpu_time=round(v["ms"] / tot_ms, 4),
                        device_copy_gbs=round(copy_gbs, 1),
                        measured_in="a separate pass of %d steps with per-kernel HIP events (%.1f ms per step; the timed "
                                    "steps run without them)" % (psteps, 1e3 * dt_prof / psteps))
            if dom == "flow_iter" and prof_levels:
                # the same class by pyramid level of the launch (0 = full resolution): three launches per level and batch,
                # credited bytes as above; the level-0 launches are most of the class's time and run at another rate
                roof["by_level"] = [dict(level=lv, launches=d["launches"], avg_launch_us=round(1e3 * d["ms"] / d["launches"], 1),
                                         share_of_class=round(d["ms"] / v["ms"], 4),
                                         achieved=round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
                                         frac=round(d["bytes"] / (d["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                                    for lv, d in sorted(prof_levels.items())]
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            cpu = cpu_baseline(chunks[args.warmup % 2][:args.cpu_frames].cpu().numpy())
        last = results[-1]
        kept = int(np.sum(last["keep"]))
        thr = float(eng.params.flow_threshold)
        margins = [hostmath.flow_margin(r["flows"], thr) for r in results]
        line = dict(metric="frames/sec (whole node) at 1080p", value=round(n_total * args.steps / dt, 2), unit="frames/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 3),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8 (flow metric in f32/f64)",
                    data="synthetic",
                    config=dict(workload=workload, frames_per_gpu=F, sharding="frame-range x%d" % world if world > 1 else "none",
                                kept_frames_last_step=kept, flow_threshold=thr,
                                min_flow_margin=round(min(m["min_margin"] for m in margins), 6),
                                frames_within_1e_3_of_threshold=sum(m["frames_within_rel"] for m in margins)),
                    roofline=roof, cpu_baseline=cpu, kernels=kernels)
        if coll:
            line["collectives"] = coll
        if strong:
            line["strong_1000"] = strong
        if world == 1 and not args.no_modes:
            try:
                line["throughput_pinned_host"], line["throughput_from_avi"] = throughput_modes(chunks, H, W)
            except Exception as e:                      # noqa: BLE001 - the extra figures never take the line down
                line["throughput_modes_error"] = repr(e)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

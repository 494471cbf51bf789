#!/usr/bin/env python3
"""bench.py -- frames/s of the damaged-pixel detection path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the whole path (ipx_detect_chunk's stages, ip.py:526-632) over one chunk of
synthetic 1920x1080 uint8 frames already resident in HBM, results copied back and turned into the
reference's four float64 arrays.  N = 1: BASELINE.json configs[1] ("1 MI355X, 1920x1080 synthetic
video", processed in the reference's 1000-frame chunks).  N > 1 (launched by torch.distributed.run,
one rank per GPU): configs[2], a chunk of N x 1000 frames is sharded by frame range, 1000 frames per GPU
(weak scaling), with the persistence / static-map / result exchanges over RCCL inside the step.

Rank 0 prints ONE JSON line.  `roofline` is measured live: libipx's built-in profiler records HIP
events on the launch stream around every kernel of the timed steps; the dominant kernel class (most
GPU time) is priced as algorithmic bytes / time against the 8 TB/s HBM peak.  `cpu_baseline` times
the CPU oracle (oracle/, a port: the reference itself needs cv2 + numba, absent from this image) on
a bounded sample of the same frames, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU per step (default 1000)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--motion", action="store_true", help="camera-motion-heavy sequence (configs[3])")
    ap.add_argument("--cpu-frames", type=int, default=128, help="frames of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel events (roofline = null)")
    return ap.parse_args()


def load_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*_traffic.json), or None."""
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


def cpu_baseline(frames_np):
    """The oracle pipeline on the sample, timed on this host's cores (OpenMP inside the C part)."""
    from oracle import pipeline
    t0 = time.perf_counter()
    pipeline.detect_damaged_pixels(frames_np)
    dt = time.perf_counter() - t0
    omp = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = min(omp, avail) if omp > 0 else avail
    n, h, w = frames_np.shape
    return dict(value=n / dt, unit="frames/s", cores=cores, kind="port",
                sample="%d frames of the step's first chunk (%dx%d), whole path incl. flow; %.1f s" % (n, w, h, dt))


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from image_processing_amd import sharding, synth
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
    H, W = args.height, args.width
    F = args.frames or 1000                  # the same work per GPU at every N (weak scaling)
    n_total = F * world
    eng = Engine(local_rank)
    comm = sharding.make_comm()
    radius = int(eng.params.sliding_window_radius)

    # ---- synthetic chunks in HBM (two distinct ones, alternated) ---------------------------------
    speeds = [0, 0.5, 4, 8] if args.motion else None
    a, b = sharding.shard_range(n_total, rank, world)
    lo, hi = sharding.halo_range(n_total, a, b, radius)
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
        """Enqueue step i; returns a callable that completes it (copy-back + the host's float glue)."""
        fr = chunks[i % 2]
        if world == 1:
            return eng.detect_chunk_async(fr).result
        out = sharding.detect_chunk_sharded(eng, fr, a, b, n_total, comm)
        return lambda: sharding.results_to_host(out, H * W)

    def run(first, count):
        """`count` steps, software-pipelined as a chunk driver would: step k+1 is on the GPU while the host
        finishes step k.  Every step's results are produced inside the call."""
        res, pending = None, None
        for i in range(first, first + count):
            nxt = submit(i)
            if pending is not None:
                res = pending()
            pending = nxt
        if pending is not None:
            res = pending()
        return res

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, args.warmup)
    if not args.no_profile:
        eng.profile_enable(True)
        eng.profile_reset()
    fence()
    t0 = time.perf_counter()
    res = run(args.warmup, args.steps)
    fence()
    dt = time.perf_counter() - t0
    prof = {} if args.no_profile else eng.profile_read()
    if not args.no_profile:
        eng.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    copy_gbs = None
    if rank == 0:
        # measured device-copy ceiling beside the 8 TB/s spec: 2 GB read + 2 GB written per copy
        src = chunks[0]
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2.0 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del dst
    if rank == 0:
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
            ach = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            traffic = load_traffic(dom, workload)
            roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
                        bytes_per_launch=round(v["bytes"] / v["launches"]), avg_launch_us=round(1e3 * v["ms"] / v["launches"], 2),
                        launches=v["launches"], share_of_gpu_time=round(v["ms"] / tot_ms, 4),
                        device_copy_gbs=round(copy_gbs, 1))
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            cpu = cpu_baseline(chunks[args.warmup % 2][:args.cpu_frames].cpu().numpy())
        kept = int(np.sum(res["keep"]))
        line = dict(metric="frames/sec (whole node) at 1080p", value=round(n_total * args.steps / dt, 2), unit="frames/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * dt / args.steps, 3),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="u8 (flow metric in f32/f64)",
                    data="synthetic",
                    config=dict(workload=workload, frames_per_gpu=F, sharding="frame-range x%d" % world if world > 1 else "none",
                                kept_frames_last_step=kept),
                    roofline=roof, cpu_baseline=cpu, kernels=kernels)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

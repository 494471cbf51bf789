#!/usr/bin/env python3
"""One stage of the path alone on a resident 1080p chunk (for rocprofv3 --pmc passes and quick A/B timing):
    python tools/bench_stage.py detect|temporal|flow [frames] [reps]
Prints the built-in profiler's per-kernel table for the stage."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine
    stage = sys.argv[1] if len(sys.argv) > 1 else "detect"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    H, W = 1080, 1920
    dev = torch.device("cuda", 0)
    eng = Engine(0)
    fr = synth.chunk_torch(2, n, H, W, dev)
    if os.environ.get("IPX_BRIGHT") == "1":   # an all-bright scene: every pixel's background is above brightness_threshold
        fr = (fr.to(torch.int16) + 130).clamp_(0, 255).to(torch.uint8)   # (detect's row pre-test then never skips the outlier cut)
    keep = torch.ones(n, dtype=torch.uint8, device=dev)
    if stage == "flow":                       # the motion test alone: wall time per call (events on the launch stream)
        eng.flow_metric(fr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(reps):
            e0.record()
            eng.flow_metric(fr)
            e1.record()
            torch.cuda.synchronize()
            print("flow_metric(%d frames): %.3f ms" % (n, e0.elapsed_time(e1)), flush=True)
        eng.profile_enable(True)
        eng.profile_reset()
        eng.flow_metric(fr)
        for k, v in sorted(eng.profile_read().items(), key=lambda kv: -kv[1]["ms"]):
            print("  %-12s launches %3d  total %8.3f ms" % (k, v["launches"], v["ms"]))
        return
    raw, bright, dense = eng.detect_raw(fr, keep)
    eng.remove_bright_regions(bright, raw, H, W, out=raw)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    eng.profile_reset()
    for _ in range(reps):
        if stage == "detect":
            eng.detect_raw(fr, keep)
        else:
            pers = eng.persistent_mask(raw, None, H, W)
            maps = eng.static_maps(fr, raw, pers, keep)
            stat = eng.static_from_maps(maps)
            eng.finalize(fr, raw, pers, stat, keep, want_bits=True)
    for k, v in sorted(eng.profile_read().items(), key=lambda kv: -kv[1]["ms"]):
        print("%-18s launches %3d  avg %9.1f us  %7.1f GB/s algorithmic" % (k, v["launches"], 1e3 * v["ms"] / v["launches"],
                                                                         v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] else 0))


if __name__ == "__main__":
    main()

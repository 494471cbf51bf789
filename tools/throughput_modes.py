#!/usr/bin/env python3
"""The two throughput figures beside bench.py's HBM-resident `value`:
  (ii)  chunks start in pinned host memory: upload on a side stream overlapped with the previous chunk;
  (iii) chunks start in an uncompressed AVI on disk: image_processing.main (decode thread + the above).
Run on the GPU box: python tools/throughput_modes.py [frames] [chunks]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import image_processing as ip
    from image_processing_amd import avi, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    H, W = 1080, 1920
    dev = torch.device("cuda", 0)
    base = [synth.chunk_torch(11 + c, n, H, W, dev) for c in range(2)]
    pinned = []
    for b in base:
        p = torch.empty(b.shape, dtype=torch.uint8, pin_memory=True)
        p.copy_(b)
        pinned.append(p)
    torch.cuda.synchronize()
    kw = dict(plot=False, show_plots=False, save_plots=False, output_folder="results", visualise_damaged_pixels_plot=True,
              plot_damaged_pixels_plot=True, plot_heatmap_plot=True, params=None)

    def run_pinned(count):
        pending = None
        for i in range(count):
            done = ip._submit_chunk(([None] * n, pinned[i % 2]), i, kw)
            if pending is not None:
                pending()
            pending = done
        pending()
    run_pinned(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_pinned(chunks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("(ii) pinned host -> HBM overlapped: %.0f frames/s (%d chunks of %d frames, %.1f GB/s of uploads)"
          % (chunks * n / dt, chunks, n, chunks * n * H * W / dt / 1e9))
    fn = os.path.join("/tmp", "ipx_bench_%d.avi" % os.getpid())
    host = [p.numpy() for p in pinned]
    n = min(n, int(3.9e9 / (H * W)) // chunks)            # a RIFF AVI holds 4 GB
    t0 = time.perf_counter()
    avi.write_avi(fn, [host[c % 2][i] for c in range(chunks) for i in range(n)], fps=25.0)
    print("wrote %s: %.1f GB in %.1f s" % (fn, os.path.getsize(fn) / 1e9, time.perf_counter() - t0))
    try:
        t0 = time.perf_counter()
        res = ip.main(fn, average_time=1.0, max_chunks=None, step_size=n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("(iii) uncompressed AVI -> decode thread -> HBM: %.0f frames/s (%d frames, %d averaging windows)"
              % (chunks * n / dt, chunks * n, len(res["averages_counts"])))
    finally:
        os.remove(fn)


if __name__ == "__main__":
    main()

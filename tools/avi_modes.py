#!/usr/bin/env python3
"""A/B of the chunk driver's input side on one box: python tools/avi_modes.py [frames_per_chunk] [chunks]
Times image_processing.main on an uncompressed 8-bit 1080p AVI (page cache) with the decode thread sending every piece
on to HBM as it is read (the default) and with the driver uploading the finished chunk, alternately, and prints whole-run
and steady-state (delivery of chunk 1 to delivery of the last) frames/s."""
import contextlib
import io
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import image_processing as ip
    from image_processing_amd import avi, synth
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 458
    nch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda", 0)
    fr = synth.chunk_torch(2, per, 1080, 1920, dev).cpu().numpy()
    fd, fn = tempfile.mkstemp(suffix=".avi", prefix="ipx_ab_")
    os.close(fd)
    try:
        avi.write_avi(fn, [fr[i] for _ in range(nch) for i in range(per)], fps=25.0)

        def one():
            ip._TIMELINE = tl = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                ip.main(fn, average_time=1.0, max_chunks=None, step_size=per)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ip._TIMELINE = None
            tl.sort()
            gaps = [round(1e3 * (tl[i + 1][2] - tl[i][2]), 1) for i in range(len(tl) - 1)]
            return nch * per / dt, (len(tl) - 1) * per / (tl[-1][2] - tl[0][2]), round(1e3 * (tl[0][2] - t0), 1), gaps
        for rnd in range(3):
            for mode in (True, False):
                ip._UPLOAD_IN_DECODER = mode
                whole, steady, first_ms, gaps = one()
                print("upload in decoder %-5s: whole %.0f frames/s, steady %.0f frames/s, first chunk delivered after %s ms, then %s" %
                      (mode, whole, steady, first_ms, gaps), flush=True)
    finally:
        os.remove(fn)


if __name__ == "__main__":
    main()

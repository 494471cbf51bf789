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
    nums = [a for a in sys.argv[1:] if a.isdigit()]
    per = int(nums[0]) if len(nums) > 0 else 458
    nch = int(nums[1]) if len(nums) > 1 else 4
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
        if "trace" in sys.argv:                          # where an odd chunk's extra milliseconds go: decode, submit, wait
            ev = []
            real_load, real_submit = ip._load_chunk, ip._submit_chunk

            import threading

            gpu_ev = []                                  # (name, chunk, event): GPU-side times against `base`

            def load(video, start, end, *a, **k):
                t = time.perf_counter(); r = real_load(video, start, end, *a, **k); ev.append(("decode", start, t, time.perf_counter()))
                if getattr(ip._TLS, "decode_stream", None) is not None:
                    e = torch.cuda.Event(enable_timing=True); e.record(ip._TLS.decode_stream); gpu_ev.append(("upload done", start, e))
                if getattr(r, "uploaded", None) is not None and "watch" in sys.argv:   # when do the chunk's bytes arrive in HBM?

                    def watch(e=r.uploaded, t=time.perf_counter()):
                        e.synchronize(); ev.append(("arrived", start, t, time.perf_counter()))
                    threading.Thread(target=watch, daemon=True).start()
                return r

            def submit(chunk, idx, kw):
                e0 = torch.cuda.Event(enable_timing=True); e0.record(); gpu_ev.append(("compute may start", idx, e0))
                t = time.perf_counter(); done = real_submit(chunk, idx, kw); ev.append(("submit", idx, t, time.perf_counter()))
                e1 = torch.cuda.Event(enable_timing=True); e1.record(); gpu_ev.append(("compute done", idx, e1))

                def fin():
                    t1 = time.perf_counter(); r = done(); ev.append(("finish", idx, t1, time.perf_counter())); return r
                return fin
            ip._load_chunk, ip._submit_chunk = load, submit
            from image_processing_amd import engine as _engine

            def timed_method(cls, name):
                real = getattr(cls, name)

                def wrapper(self, *a, **k):
                    t = time.perf_counter(); r = real(self, *a, **k); ev.append((name, 0, t, time.perf_counter())); return r
                setattr(cls, name, wrapper)
            timed_method(_engine.Engine, "unpack_dib")
            timed_method(_engine.Engine, "detect_chunk_async")
            if "cold" in sys.argv:                       # the first pass of the process: what a one-video job sees
                t00 = time.perf_counter()
                whole, steady, first_ms, gaps = one()
                print("first pass of the process: whole %.0f frames/s, first chunk after %s ms, then %s" % (whole, first_ms, gaps))
                for name, k, a, b in sorted(ev, key=lambda e: e[2]):
                    print("    %-7s %6d  %8.1f -> %8.1f ms (%6.1f)" % (name, k, 1e3 * (a - t00), 1e3 * (b - t00), 1e3 * (b - a)))
            else:
                one()
            for rnd in range(8):
                del ev[:]
                del gpu_ev[:]
                base = torch.cuda.Event(enable_timing=True); base.record(); torch.cuda.synchronize()
                t00 = time.perf_counter()
                whole, steady, first_ms, gaps = one()
                torch.cuda.synchronize()
                print("  GPU clock: " + "; ".join("%s %d at %.1f" % (n, k, base.elapsed_time(e)) for n, k, e in gpu_ev))
                print("pass %d: steady %.0f frames/s, gaps %s" % (rnd, steady, gaps))
                for name, k, a, b in sorted(ev, key=lambda e: e[2]):
                    print("    %-7s %6d  %8.1f -> %8.1f ms (%6.1f)" % (name, k, 1e3 * (a - t00), 1e3 * (b - t00), 1e3 * (b - a)))
            return
        if "threads" in sys.argv:                        # sweep of the reader threads / piece size instead of the A/B
            one()
            for rnd in range(2):
                for th, piece in ((4, 64), (8, 64), (12, 64), (16, 64), (8, 32), (8, 128), (16, 32)):
                    ip._DECODE_THREADS, ip._UPLOAD_PIECE = th, piece << 20
                    whole, steady, first_ms, gaps = one()
                    print("reader threads %2d, pieces of %3d MB: whole %.0f frames/s, steady %.0f frames/s, first chunk after %s ms, then %s" %
                          (th, piece, whole, steady, first_ms, gaps), flush=True)
            return
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

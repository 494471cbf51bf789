#!/usr/bin/env python3
"""Host-to-device copy rate of a 2 GB pinned buffer split over 1, 2, 4 streams, alone and under the flow stage
(what the chunk driver's upload competes with).  python tools/h2d_probe.py"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine
    dev = torch.device("cuda", 0)
    n = 2 << 30
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    host.numpy()[:] = 7
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    eng = Engine(0)
    fr = synth.chunk_torch(2, 513, 1080, 1920, dev)
    eng.flow_metric(fr)
    torch.cuda.synchronize()

    def copy(ns, piece=None):
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if piece is None:
            per = n // ns
            for i, s in enumerate(streams):
                with torch.cuda.stream(s):
                    dst[i * per:(i + 1) * per].copy_(host[i * per:(i + 1) * per], non_blocking=True)
        else:
            for k, a in enumerate(range(0, n, piece)):
                with torch.cuda.stream(streams[k % ns]):
                    dst[a:a + piece].copy_(host[a:a + piece], non_blocking=True)
        for s in streams:
            s.synchronize()
        return n / (time.perf_counter() - t0) / 1e9

    for busy in (False, True):
        for ns, piece in ((1, None), (2, None), (4, None), (2, 64 << 20), (4, 64 << 20)):
            stop = threading.Event()
            if busy:
                def work():
                    torch.cuda.set_device(dev)
                    while not stop.is_set():
                        eng.flow_metric(fr)
                        torch.cuda.current_stream().synchronize()
                th = threading.Thread(target=work)
                th.start()
                time.sleep(0.1)
            r = [copy(ns, piece) for _ in range(3)]
            stop.set()
            if busy:
                th.join()
            print("%s  streams %d  pieces %s: %s GB/s" % ("under the flow stage" if busy else "idle GPU            ", ns,
                                                         "whole" if piece is None else "%d MB" % (piece >> 20), ", ".join("%.1f" % x for x in r)), flush=True)


if __name__ == "__main__":
    main()

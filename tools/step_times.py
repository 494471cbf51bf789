#!/usr/bin/env python3
"""python tools/step_times.py [steps]: wall time of every step of bench.py's pipelined loop in one process (1 GPU, 1080p, 1000-frame
chunks), to see how many warm-up steps the steady state needs."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from image_processing_amd import engine, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
eng = engine.Engine()
chunks = [synth.chunk_torch(1000 * c + 2, 1000, 1080, 1920, dev) for c in range(2)]
torch.cuda.synchronize()
pending, t_prev, out = None, time.perf_counter(), []
for i in range(steps + 1):
    nxt = eng.detect_chunk_async(chunks[i % 2]).result if i < steps else None
    if pending is not None:
        pending()
        t = time.perf_counter()
        out.append(1e3 * (t - t_prev))
        t_prev = t
    else:
        t_prev = time.perf_counter()
    pending = nxt
print("ms per completed step:", " ".join("%.1f" % x for x in out))

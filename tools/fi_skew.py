#!/usr/bin/env python3
"""How far apart the workgroups that share an expansion run (debug build of tools/fi_stamps.py --build): start and end
time of every workgroup of the last level-0 launch of a 512-pair batch, from s_memtime (100 MHz).
  python tools/fi_skew.py        on the GPU box, after `python tools/fi_stamps.py --build` here"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["IPX_LIB"] = os.path.join(ROOT, "image_processing_amd", "lib_stamps", "libipx.so")


def main():
    import torch
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine
    from image_processing_amd._ipx import lib
    eng = Engine(0)
    fr = synth.chunk_torch(3, 513, 1080, 1920, torch.device("cuda", 0))
    eng.flow_metric(fr)
    eng.flow_metric(fr)
    torch.cuda.synchronize()
    strips, ppx, G = 17, 64, int(os.environ.get("IPX_FI_STRIP_GROUP", "3"))
    n = 8 * ppx * strips
    buf = (C.c_ulonglong * (2 * n))()
    lib.ipx_debug_fi_wg_times.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert lib.ipx_debug_fi_wg_times(buf, n) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2, n).astype(np.int64)
    # s_memtime counts shader-clock cycles of the workgroup's XCD (counters of different XCDs are unrelated): everything
    # below compares workgroups of one XCD, in units of a step (duration / number of steps)
    steps = (1080 + 3) // 4
    step_cycles = float(np.median(t[1] - t[0])) / steps
    start, end = t[0] / step_cycles, t[1] / step_cycles           # in steps
    print("workgroup duration: median %.0f cycles = %d steps of %.0f cycles" % (np.median(t[1] - t[0]), steps, step_cycles))
    x0 = np.arange(0, 8 * 70, 8)                                  # XCD 0, its first 70 workgroups
    print("XCD 0, first 70 workgroups: start (steps after the first)", np.round(start[x0] - start[x0].min(), 1).tolist())
    print("XCD 0, first 70 workgroups: end - start (steps)", np.round((end - start)[x0], 1).tolist())
    # workgroup id -> (xcd, pair in xcd, strip) as flow_iter_kernel maps it
    dstart, dend, ddur = [], [], []
    for xcd in range(8):
        where = {}
        for k in range(ppx * strips):
            gi, rem = divmod(k, G * ppx)
            gs = min(G, strips - gi * G)
            pl, ts = divmod(rem, gs)
            where[(pl, gi * G + ts)] = 8 * k + xcd
        for (pl, s), i in where.items():
            j = where.get((pl + 1, s))
            if j is not None:
                dstart.append(start[j] - start[i]); dend.append(end[j] - end[i]); ddur.append((end[j] - start[j]) - (end[i] - start[i]))
    for name, d in (("start", dstart), ("end", dend), ("duration", ddur)):
        d = np.abs(np.array(d))
        print("pair p+1 against pair p, same strip: |%s difference| median %.1f steps, 90 %% %.1f, max %.1f" % (name, np.median(d), np.percentile(d, 90), d.max()))
    print("(the L2 holds about 3 steps of the resident workgroups' lines)")


if __name__ == "__main__":
    main()

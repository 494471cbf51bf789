#!/usr/bin/env python3
"""In-kernel phase timing of flow_iter_kernel (shader-clock stamps, debug build only).

  python tools/fi_stamps.py --build     here (CPU): compiles image_processing_amd/lib_stamps/libipx.so with -DIPX_FI_STAMPS
  python tools/fi_stamps.py             on the GPU box: runs one 1080p batch and prints cycles per step, wave and phase

Producer waves (4-7): 0 = finish + store the first of the thread's two matrix rows (includes waiting for its loads),
1 = issue the loads that replace it, 2 = finish + store the second row, 3 = issue its replacement, 4 = wait at the
step barrier.  Consumer waves (0-3): 1 = B (vertical running sums), 2 = wait for the other consumers, 3 = C
(horizontal sums + solve + store), 4 = wait at the step barrier.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "image_processing_amd", "lib_stamps")
LIB = os.path.join(LIBDIR, "libipx.so")


def build():
    from image_processing_amd import build as b
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for s in b.SOURCES:
        obj = os.path.join(LIBDIR, s.replace(".hip", ".o"))
        subprocess.check_call([b._hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get(s, []) + ["-DIPX_FI_STAMPS", "-c", os.path.join(b.CSRC, s), "-o", obj])
        objs.append(obj)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print(LIB)


def run():
    os.environ["IPX_LIB"] = LIB
    import torch
    from image_processing_amd import synth
    from image_processing_amd.engine import Engine
    from image_processing_amd._ipx import lib
    eng = Engine(0)
    fr = synth.chunk_torch(3, 65, 1080, 1920, torch.device("cuda", 0))
    eng.flow_metric(fr)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.ipx_debug_fi_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    assert lib.ipx_debug_fi_stamps(buf, 1) == 0
    eng.flow_metric(fr)
    torch.cuda.synchronize()
    assert lib.ipx_debug_fi_stamps(buf, 1) == 0
    print("wave  steps   " + "".join("%9s" % ("ph%d" % i) for i in range(5)) + "    total   (cycles per step)")
    for w in range(8):
        row = [buf[w * 8 + i] for i in range(8)]
        steps = max(row[7], 1)
        ph = [row[i] / steps for i in range(5)]
        print("%4d %6d   " % (w, steps) + "".join("%9.0f" % v for v in ph) + "%9.0f" % sum(ph))


if __name__ == "__main__":
    build() if "--build" in sys.argv else run()

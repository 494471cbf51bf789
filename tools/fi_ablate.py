#!/usr/bin/env python3
"""Ablation builds of flow_iter_kernel: which resource sets the time of a step (timing only, results are garbage).

  python tools/fi_ablate.py --build [bits ...]   here (CPU): image_processing_amd/lib_abl<bits>/libipx.so, flow_iter.hip with -DIPX_FI_ABLATE=<bits>
  python tools/fi_ablate.py [bits ...]           on the GPU box: tools/bench_stage.py flow 257 3 on every variant, one process each

Bits (csrc/flow_iter.h): 1 = consumer waves idle, 2 = second expansion sampled from the first frame (L1 hits),
4 = producers add up what they loaded instead of the matrix arithmetic; further bits are listed there.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEFAULT = [0, 1, 2, 4, 5, 7]


def libdir(bits):
    return os.path.join(ROOT, "image_processing_amd", "lib_abl%d" % bits)


def build(variants):
    from image_processing_amd import build as b
    b.build_lib()
    procs = []
    for v in variants:
        d = libdir(v)
        os.makedirs(d, exist_ok=True)
        obj = os.path.join(d, "flow_iter.o")
        procs.append((v, subprocess.Popen([b._hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get("flow_iter.hip", []) +
                                          ["-DIPX_FI_ABLATE=%d" % v, "-c", os.path.join(b.CSRC, "flow_iter.hip"), "-o", obj])))
    for v, p in procs:
        if p.wait() != 0:
            raise SystemExit("variant %d failed to compile" % v)
        objs = [os.path.join(b.LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES if s != "flow_iter.hip"] + [os.path.join(libdir(v), "flow_iter.o")]
        subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(libdir(v), "libipx.so")] + objs)
        print(os.path.join(libdir(v), "libipx.so"))


def run(variants, frames):
    for v in variants:
        env = dict(os.environ, IPX_LIB=os.path.join(libdir(v), "libipx.so"))
        print("== IPX_FI_ABLATE=%d" % v, flush=True)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_stage.py"), "flow", str(frames), "3"], env=env, timeout=300)
        if r.returncode != 0:
            raise SystemExit("variant %d: exit code %d" % (v, r.returncode))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    frames = 257
    for a in sys.argv[1:]:
        if a.startswith("--frames="):
            frames = int(a.split("=")[1])
    variants = [int(a) for a in args] or DEFAULT
    build(variants) if "--build" in sys.argv else run(variants, frames)

#!/usr/bin/env python3
"""Variant builds of the flow stage for A/B timing on one box (tools/ab.sh, tools/ab4.sh):

  python tools/variant.py <name> [-DMACRO=value ...]   here (CPU): image_processing_amd/lib_<name>/libipx.so,
                                                       flow.hip and flow_iter.hip compiled with the extra flags
  VARIANT_SRCS=ccl.hip,detect.hip python tools/variant.py ...   other translation units instead
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from image_processing_amd import build as b
    name, extra = sys.argv[1], sys.argv[2:]
    b.build_lib()
    d = os.path.join(ROOT, "image_processing_amd", "lib_" + name)
    os.makedirs(d, exist_ok=True)
    mine = os.environ.get("VARIANT_SRCS", "flow.hip,flow_iter.hip").split(",")
    procs = [subprocess.Popen([b._hipcc()] + b.FLAGS + b.EXTRA_FLAGS.get(s, []) + extra +
                              ["-c", os.path.join(b.CSRC, s), "-o", os.path.join(d, s.replace(".hip", ".o"))]) for s in mine]
    if any(p.wait() != 0 for p in procs):
        raise SystemExit("compile failed")
    objs = [os.path.join(d if s in mine else b.LIBDIR, s.replace(".hip", ".o")) for s in b.SOURCES]
    out = os.path.join(d, "libipx.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-lpthread"])
    print(out)


if __name__ == "__main__":
    main()

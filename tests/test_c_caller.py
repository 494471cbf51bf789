"""include/ipx.h from plain C: tests/c_abi/group_example.c is compiled as pedantic C99 against the header and linked
against libipx.so (CPU: the header is valid C and every symbol it uses resolves); on the GPU box the program runs:
ipx_detect_chunk vs ipx_group_detect_chunk (three ranks on device 0) with nothing but C between caller and library."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "group_example.c")
LIBDIR = os.path.join(ROOT, "image_processing_amd", "lib")


def _build(out):
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", LIBDIR, "-lipx", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return out


def test_header_is_c_and_the_example_links(tmp_path):
    _build(str(tmp_path / "group_example"))


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 3])
def test_c_program_runs_group_equal_to_single_context(tmp_path, ranks):
    sys.path.insert(0, ROOT)
    from image_processing_amd import hostmath
    exe = _build(str(tmp_path / "group_example"))
    tab, tdim = hostmath.pow2_table(20)                    # ip.py:173 max_cluster_size default
    tfile = tmp_path / "pow2.bin"
    np.ascontiguousarray(tab, dtype=np.float64).tofile(str(tfile))
    r = subprocess.run([exe, str(tfile), str(tdim), str(ranks)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok:"), r.stdout
    kept = int(r.stdout.split("frames,")[1].split("kept")[0])
    assert 0 < kept < 24                                   # the jump at frame 13 is discarded, the rest is kept

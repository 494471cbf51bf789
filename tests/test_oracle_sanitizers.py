"""`make -C oracle asan`: the checker's own C code -- oracle/oracle.c, every entry point once on small and
awkward shapes -- under AddressSanitizer + UndefinedBehaviorSanitizer, and the C99 caller of include/ipx.h compiled with
the same flags.  Host code only; nothing here touches a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc and make")
def test_oracle_c_is_clean_under_asan_and_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "oracle asan driver: ok" in r.stdout, r.stdout[-4000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-4000:]

"""CPU-only checks of the boundary: libipx.so loads and exports every function include/ipx.h
declares (no compute call is made), the ctypes table covers the header, the params struct has the
header's layout and defaults, and the host float glue reproduces the reference's numbers."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ipx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ipx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from image_processing_amd import _ipx
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(_ipx.lib, n), "libipx.so does not export %s" % n
    assert sorted(_ipx.PROTOTYPES) == names, "ctypes table and include/ipx.h differ"


def test_version_error_string_and_defaults():
    from image_processing_amd import _ipx
    assert _ipx.lib.ipx_version() >= 100
    assert _ipx.lib.ipx_words_per_row(1) == 1 and _ipx.lib.ipx_words_per_row(64) == 1
    assert _ipx.lib.ipx_words_per_row(65) == 2 and _ipx.lib.ipx_words_per_row(1920) == 30
    p = _ipx.default_params()                              # ip.py:167-177 and the hard-coded constants
    assert (p.consecutive_threshold, p.sliding_window_radius, p.min_cluster_size, p.max_cluster_size) == (2, 3, 5, 20)
    assert (p.min_valid_frames, p.circularity_size_threshold, p.estimate_threshold) == (10, 10, 170)
    assert (p.brightness_threshold, p.flow_threshold, p.static_threshold, p.min_circularity, p.pixel_std_coeff) == \
        (170.0, 2.0, 50.0, 0.1, 1.0)
    assert C.sizeof(_ipx.Params) == 8 * 4 + 5 * 8
    # argument errors are reported through the status code + message, without touching a GPU
    assert _ipx.lib.ipx_device_count(None) == -1
    assert b"null pointer" in _ipx.lib.ipx_last_error()


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from image_processing_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image_processing_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    for f in ("image_processing.py",):
        path = os.path.join(ROOT, f)
        if os.path.exists(path):
            assert not re.search(r"^\s*(from|import)\s+oracle\b", open(path).read(), flags=re.M)


def test_bright_area_estimate_matches_oracle_formula():
    from image_processing_amd import hostmath
    from oracle import stages as S
    rng = np.random.default_rng(0)
    for _ in range(200):
        h, w = 9, 11
        frame = rng.integers(120, 256, (h, w)).astype(np.uint8)
        mask = rng.random((h, w)) < 0.2
        bt = int(rng.choice([100, 170, 254]))
        exp = S.find_bright_area_estimates([frame.astype(np.float64)], [mask], bt)[0]
        got = hostmath.bright_area_estimate(int(mask.sum()), int((mask & (frame < 170)).sum()),
                                            int((mask & (frame > bt)).sum()), int((frame < 170).sum()),
                                            int((frame > bt).sum()), h * w)
        assert (math.isnan(got) and math.isnan(exp)) or got == exp
    # Python round: half to even
    assert hostmath.bright_area_estimate(1, 1, 0, 3, 5, 8) == round(1 / 2 * 5)


def test_chunk_results_nan_rules():
    from image_processing_amd import hostmath
    keep = np.array([1, 0, 1], np.uint8)
    counts = np.array([[3, 3, 0, 50, 0], [0, 0, 0, 0, 0], [0, 0, 0, 64, 0]], np.int64)
    cluster = np.array([[2, 11, 1234], [0, 0, 0], [0, 0, 0]], np.int64)
    total, cc, cs, cb = hostmath.chunk_results(keep, counts, cluster, 64)
    assert total[0] == 3 and np.isnan(total[1]) and total[2] == 0          # ip.py:359-362
    assert cc.tolist() == [2, 0, 0] and cs[0] == 5.5 and cb[0] == 1234 / 11
    assert cs[1] == 0 and cb[1] == 0                                       # discarded: zeros, ip.py:335
    assert cs[2] == 0 and np.isnan(cb[2])                                  # kept, no cluster: nan, ip.py:680


def test_pow2_table_is_python_pow():
    from image_processing_amd import hostmath
    tab, dim = hostmath.pow2_table(20)
    assert dim >= 48 and tab.shape == (dim * dim,)
    for a, d in [(0, 0), (3, 7), (10, 4), (20, 20)]:
        assert tab[a * dim + d] == (a * 1.0 + d * hostmath.SQRT2_F32) ** 2


def test_no_64bit_shift_by_last_allocated_vgpr():
    """gfx950 mis-executes a 64-bit vector shift whose amount is the last VGPR of the allocation (measured:
    tools/hwtest/seg_loop.hip); the build refuses such kernels, and so does this test."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_shift64.py")
    spec = importlib.util.spec_from_file_location("check_shift64", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sample = """
_Zbad:
\tv_lshlrev_b64 v[12:13], v15, -1
\ts_endpgm
\t.amdhsa_kernel _Zbad
\t\t.amdhsa_next_free_vgpr 16
_Zgood:
\tv_lshlrev_b64 v[12:13], v15, -1
\ts_endpgm
\t.amdhsa_kernel _Zgood
\t\t.amdhsa_next_free_vgpr 17
"""
    assert [b[0] for b in mod.check_asm(sample)] == ["_Zbad"]
    # the same scan refuses kernels whose register arrays ended up in private (scratch) memory
    spills = """
	.amdhsa_kernel _Zspills
		.amdhsa_private_segment_fixed_size 64
	.amdhsa_kernel _Zclean
		.amdhsa_private_segment_fixed_size 0
	.amdhsa_kernel _ZN12_GLOBAL__N_117persistent_kernelILi24ELb1EEEvPKmPKhimiiPm
		.amdhsa_private_segment_fixed_size 24
"""
    assert mod.check_scratch(spills) == [("_Zspills", 64)]
    # and kernels with a wide buffer store through a register scalar offset that no wait state follows (on gfx950 a write to
    # the data registers right behind such a store reaches memory: csrc/flow_iter.h, ipx_buffer_store_b128)
    stores = """
_Zstores:
\tbuffer_store_dwordx4 v[4:7], v0, s[44:47], s52 offen nt
\tv_mul_f64 v[4:5], v[8:9], s[30:31]
\tbuffer_store_dwordx4 v[4:7], v0, s[44:47], 0 offen
\tv_mov_b32 v4, 0
\tbuffer_store_dwordx4 v[8:11], v0, s[44:47], s5 offen
\t; sched_barrier mask(0x00000000)
\ts_nop 1
\tbuffer_store_dwordx2 v[4:5], v0, s[44:47], s5 offen
\ts_endpgm
\t.amdhsa_kernel _Zstores
\t\t.amdhsa_next_free_vgpr 12
"""
    assert mod.check_wide_stores(stores) == [("_Zstores", "buffer_store_dwordx4 v[4:7], v0, s[44:47], s52 offen nt")]
    assert mod.check_library() == []


def test_f32_quotients_via_f64_reciprocal_are_exact():
    """csrc/detect.hip replaces the f32 divisions S/n, Si/c and bg/255 by RN32(f64(a) * RN64(1/b)):
    every S/n and Si/c a window of up to 16 neighbours can produce, and bg/255 over those quotients
    plus a random sample of floats in [0, 256)."""
    quot = []
    for n in range(1, 17):
        a = np.arange(0, 255 * n + 1)
        ieee = a.astype(np.float32) / np.float32(n)
        fast = (a.astype(np.float64) * (1.0 / n)).astype(np.float32)
        assert np.array_equal(ieee, fast), n
        quot.append(ieee)
    rng = np.random.default_rng(5)
    bg = np.concatenate(quot + [rng.uniform(0, 256, 2_000_000).astype(np.float32),
                                np.ldexp(rng.integers(1, 1 << 24, 500_000).astype(np.float32), -16)])
    assert np.array_equal(bg / np.float32(255.0), (bg.astype(np.float64) * (1.0 / 255.0)).astype(np.float32))


def test_f32_quotients_by_residual_correction_are_exact():
    """csrc/detect.hip (div_small) forms Si/c and bg/255 in f32 as q = RN(a*y), q + RN(a - b*q)*y with
    y = RN32(1/b) and the residual from an fma.  Walked through exactly (rationals) for every Si/c a window of
    up to 16 neighbours can produce and for every such quotient divided by 255: the result is the IEEE f32 quotient."""
    from fractions import Fraction as F
    import math

    def rn32(fr):
        if fr == 0:
            return F(0)
        sgn = 1 if fr > 0 else -1
        a = abs(fr)
        e = math.floor(math.log2(float(a)))
        while F(2) ** e > a:
            e -= 1
        while F(2) ** (e + 1) <= a:
            e += 1
        q = F(2) ** (e - 23)
        k = a / q
        fl = k.numerator // k.denominator
        rem = k - fl
        if rem > F(1, 2) or (rem == F(1, 2) and fl % 2 == 1):
            fl += 1
        return sgn * fl * q

    def div_small(a, b, y):
        q = rn32(a * y)
        r = rn32(a - b * q)                              # fma: one rounding of the exact residual
        return rn32(q + r * y)

    bgs = set()
    for c in range(1, 17):
        y = rn32(F(1, c))
        assert y == F(float(np.float32(1.0) / np.float32(c)))
        for si in range(0, 255 * c + 1):
            exact = rn32(F(si, c))
            assert div_small(F(si), F(c), y) == exact, (si, c)
            bgs.add(exact)
    y = rn32(F(1, 255))
    for bg in bgs:
        assert div_small(bg, F(255), y) == rn32(bg / 255), float(bg)

#!/usr/bin/env python3
"""Build-time guards against gfx950 / hipcc 7.2 problems that were measured here (the first in round 1, tools/hwtest/seg_loop.hip;
check_wide_stores and check_scratch further down say what they guard):

a 64-bit vector shift whose shift-amount VGPR is the LAST register of the wave's allocation (the kernel's
next_free_vgpr is a multiple of the allocation granule and the amount sits in next_free_vgpr - 1) sometimes
shifts by the low bits of a register of the neighbouring allocation.  LLVM knows this as gfx90a's
Shift64HighRegBug; hipcc 7.2 does not apply the workaround to gfx950.

    python tools/check_shift64.py            compiles every csrc/*.hip to ISA and lists offending kernels (exit 1)

A kernel is cured by keeping a register allocated that the code does not use, so that the shift amount cannot be
the allocation's last register: IPX_VGPR_PAD_SAFE() (ipx_internal.h; v62 -> a 64-register allocation whose last
register the compiler never hands out), or IPX_VGPR_PAD(n) with n = the kernel's next_free_vgpr printed here.

What the characterisation covers and what it does not (measured with tools/hwtest/seg_loop.hip only): the three
opcodes below with a VGPR shift amount.  NOT tested: shift amounts held in SGPRs or given as constants, the
64-bit shift folded into v_lshl_add_u64, other allocation granules than 8, and other opcodes that read a 64-bit
pair next to the allocation's end.  The scan therefore guards the failure that was observed, not a class.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRANULE = 8


def kernels_of(asm_text):
    """yield (name, body_lines, next_free_vgpr)"""
    cur, body, out = None, [], []
    free = {}
    for line in asm_text.splitlines():
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            if cur:
                out.append((cur, body))
            cur, body = m.group(1), []
            continue
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            kname = m.group(1)
            continue
        m = re.match(r"\s*\.amdhsa_next_free_vgpr\s+(\d+)", line)
        if m:
            free[kname] = int(m.group(1))
            continue
        if cur:
            body.append(line)
    if cur:
        out.append((cur, body))
    for name, b in out:
        if name in free:
            yield name, b, free[name]


SHIFT = re.compile(r"\s*(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)\s+v\[\d+:\d+\],\s*v(\d+)\s*,")


def check_asm(asm_text):
    bad = []
    for name, body, nfree in kernels_of(asm_text):
        if nfree % GRANULE:
            continue
        for line in body:
            m = SHIFT.match(line)
            if m and int(m.group(2)) == nfree - 1:
                bad.append((name, nfree, line.strip()))
    return bad


# Kernels that may use private (scratch) memory: everything else must keep its arrays in registers.  A register
# array that the compiler cannot keep there (e.g. one assigned under a condition inside an unrolled loop) moves to
# scratch silently and costs tens of per cent (polyexp_kernel in round 2: +31 %).
SCRATCH_ALLOWED = ("persistent_kernel",)
PRIVATE = re.compile(r"\s*\.amdhsa_private_segment_fixed_size\s+(\d+)")


def check_scratch(asm_text):
    """[(kernel, bytes)] for kernels with a private segment that are not in SCRATCH_ALLOWED."""
    bad, kname = [], None
    for line in asm_text.splitlines():
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            kname = m.group(1)
            continue
        m = PRIVATE.match(line)
        if m and int(m.group(1)) > 0 and kname and not any(a in kname for a in SCRATCH_ALLOWED):
            bad.append((kname, int(m.group(1))))
    return bad


# A buffer store of more than 8 bytes whose scalar offset is a REGISTER: the compiler leaves no distance between it and a
# following write to its data registers (it does for a constant scalar offset), and on gfx950 the store then writes the new
# contents (round 2: polyexp_kernel stored garbage in launches of more than about one workgroup per CU).  Such a store must be
# followed by a wait state: ipx_buffer_store_b128 (csrc/flow_iter.h) issues `s_nop 1` behind it.
WIDE_STORE = re.compile(r"\s*buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)\b")


def check_wide_stores(asm_text):
    """[(kernel, line)] for wide buffer stores with a register scalar offset that no s_nop follows."""
    bad = []
    for name, body, _ in kernels_of(asm_text):
        code = [l for l in body if l.strip() and not l.strip().startswith(";")]
        for i, line in enumerate(code):
            if WIDE_STORE.match(line) and not (i + 1 < len(code) and code[i + 1].strip().startswith("s_nop")):
                bad.append((name, line.strip()))
    return bad


def compile_to_asm(src, flags, hipcc="/opt/rocm/bin/hipcc"):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc] + flags + ["-S", "--cuda-device-only", "-o", out, src],
                              stderr=subprocess.DEVNULL)
        with open(out) as f:
            return f.read()


def check_library():
    sys.path.insert(0, ROOT)
    from image_processing_amd import build as b
    flags = [f for f in b.FLAGS if f not in ("-fPIC",)]
    bad = []
    for s in b.SOURCES:
        asm = compile_to_asm(os.path.join(b.CSRC, s), flags + b.EXTRA_FLAGS.get(s, []), b._hipcc())                                  # the build's compiler
        bad += [(s,) + x for x in check_asm(asm)]
        bad += [(s, name, 0, "uses %d bytes of private (scratch) memory per lane" % nbytes) for name, nbytes in check_scratch(asm)]
        bad += [(s, name, 0, "no wait state behind a wide buffer store with a register scalar offset: " + line) for name, line in check_wide_stores(asm)]
    return bad


if __name__ == "__main__":
    bad = check_library()
    for src, name, nfree, line in bad:
        print("%s: kernel %s (next_free_vgpr %d): %s" % (src, name, nfree, line))
    print("%d offending kernel(s)" % len(bad))
    sys.exit(1 if bad else 0)

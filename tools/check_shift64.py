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

What the characterisation covers (round 3: tools/hwtest/shift_forms.hip places each operand in the last registers of a
16-VGPR allocation on purpose, 16 M lanes x 8 launches per form; output in profiles/r03m_shift64_forms.txt):

    FAILS   v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 with the AMOUNT in the last allocated VGPR and the shifted value
            NOT in a VGPR pair -- an inline constant (`v_lshlrev_b64 v[a:b], v15, -1`, the instruction of ccl.hip's
            segment loop) or an SGPR pair.  The wrong results are shifts by the low bits of what the next allocation's
            first register holds.  Wait states before the shift and loads in flight make no difference.
    ok      the same three opcodes with the amount in the last VGPR and the value in a VGPR pair (anywhere);
            the value or the destination pair in the last two VGPRs; the amount in an SGPR or a constant;
            v_lshl_add_u64 with either 64-bit operand in the last pair; the amount in the last-but-one VGPR.

The scan below flags every one of the three opcodes whose amount is the last allocated VGPR, whatever the other operand is:
a superset of the failing forms.  Not measured: allocation granules other than 8 (the scan assumes 8), other opcodes that
take a 32-bit VGPR operand beside a 64-bit constant or SGPR operand.
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
# Round 3 (tools/hwtest/wide_store_forms.hip, profiles/r03i_wide_store_forms.txt), every form with no wait state / s_nop 0 / s_nop 1:
#   buffer_store_dwordx2 (register soffset), global_store_dwordx2: no hazard;
#   buffer_store_dwordx3 / x4 with a REGISTER soffset: wrong data at 16 workgroups per CU without a wait state (0.15 % of the
#     dwords), none with one -- the 12-byte store is affected like the 16-byte one;
#   buffer_store_dwordx3 / x4 with soffset 0 and global_store_dwordx3 / x4 (the forms the compiler covers): wrong data without
#     a wait state (5.7 %), still 0.2 % with ONE, none with two (`s_nop 1`, what the compiler inserts for gfx940 and later).
# The scan therefore asks for `s_nop 1` or longer behind the register-soffset forms, x3 included.
WIDE_STORE = re.compile(r"\s*buffer_store_dwordx[34]\s+v\[\d+:\d+\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)\b")


def check_wide_stores(asm_text):
    """[(kernel, line)] for wide buffer stores with a register scalar offset that no s_nop follows."""
    bad = []
    for name, body, _ in kernels_of(asm_text):
        code = [l for l in body if l.strip() and not l.strip().startswith(";")]
        for i, line in enumerate(code):
            if not WIDE_STORE.match(line):
                continue
            nxt = code[i + 1].strip() if i + 1 < len(code) else ""
            m = re.match(r"s_nop\s+(\d+)", nxt)
            if not (m and int(m.group(1)) >= 1):
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

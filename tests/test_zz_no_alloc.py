"""ipx_remove_bright_regions_noalloc (include/ipx.h): the entry point that allocates nothing.

Called through the C ABI (ctypes) and compared bit for bit with ipx_remove_bright_regions on the same
planes, twice in a row on one context, the second call with a smaller input: the static scratch keeps the
first call's contents and nothing clears it in between.  Also: a third call with planes of another width
(other segment ids), the refusal of planes beyond the static array, and that no device memory is
allocated by the call (torch's allocator aside, HIP's free-memory figure does not move).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planes(rng, count, h, w, density):
    """(bright, damaged) bool stacks: blobs of several sizes + noise, so that components of both planes overlap."""
    bright = np.zeros((count, h, w), bool)
    dmg = rng.random((count, h, w)) < density
    for i in range(count):
        for _ in range(int(rng.integers(3, 12))):
            y, x = int(rng.integers(0, h)), int(rng.integers(0, w))
            hh, ww = int(rng.integers(1, 12)), int(rng.integers(1, 40))
            bright[i, y:y + hh, x:x + ww] = True
            if rng.random() < 0.7:                                  # a damaged streak that touches the blob
                dmg[i, max(0, y - 1):y + 1, max(0, x - 3):x + 2] = True
        bright[i] |= rng.random((h, w)) < 0.002
    return bright, dmg


def test_noalloc_matches_allocating_entry_point_on_consecutive_calls():
    import torch
    from image_processing_amd import _ipx
    from image_processing_amd.engine import Engine
    eng = Engine(0)
    rng = np.random.default_rng(77)
    try:
        assert _ipx.lib.ipx_noalloc_scratch_bytes() == 131072 * 520
        # first call large (several labelling batches inside the static array), second smaller, third another width
        for count, h, w, density in ((9, 270, 480, 0.02), (3, 96, 200, 0.05), (2, 130, 70, 0.3)):
            bright, dmg = _planes(rng, count, h, w, density)
            bb = eng.pack_mask(torch.from_numpy(bright).to(eng.device))
            db = eng.pack_mask(torch.from_numpy(dmg).to(eng.device))
            exp = eng.remove_bright_regions(bb, db, h, w)
            out = torch.full_like(db, -1)
            torch.cuda.synchronize()
            free0, _ = torch.cuda.mem_get_info(eng.device)
            eng.remove_bright_regions_noalloc(bb, db, h, w, out)
            torch.cuda.synchronize()
            free1, _ = torch.cuda.mem_get_info(eng.device)
            assert free1 == free0, "device memory was allocated or freed inside the call"
            assert bool((out == exp).all()), (count, h, w)
            kept = eng.unpack_mask(out, h, w).cpu().numpy().astype(bool)
            assert kept.any() and (dmg & ~kept).any(), "the case must both keep and remove components"
        # in place (out aliases the damaged planes), as the chunk path calls the allocating entry point
        eng.remove_bright_regions_noalloc(bb, db, h, w, db)
        assert bool((db == exp).all())
    finally:
        eng.close()


def test_noalloc_full_hd_planes_and_size_limit():
    import torch
    from image_processing_amd import _ipx
    from image_processing_amd.engine import Engine
    eng = Engine(0)
    rng = np.random.default_rng(5)
    try:
        h, w = 1080, 1920                                            # 4 planes per batch in the static array; 6 planes = 2 batches
        bright, dmg = _planes(rng, 6, h, w, 0.001)
        bb = eng.pack_mask(torch.from_numpy(bright).to(eng.device))
        db = eng.pack_mask(torch.from_numpy(dmg).to(eng.device))
        exp = eng.remove_bright_regions(bb, db, h, w)
        out = torch.empty_like(db)
        eng.remove_bright_regions_noalloc(bb, db, h, w, out)
        assert bool((out == exp).all())
        # beyond the static array: refused with a clear error, nothing launched
        big_h, big_w = 2200, 3840                                    # 2200 * 60 = 132,000 words > 131,072
        dummy = torch.zeros(8, dtype=torch.int64, device=eng.device)
        with pytest.raises(_ipx.IpxError) as ei:
            eng._enter()
            _ipx.check(_ipx.lib.ipx_remove_bright_regions_noalloc(eng._ctx, C.c_void_p(dummy.data_ptr()), C.c_void_p(dummy.data_ptr()),
                                                                  1, big_h, big_w, C.c_void_p(dummy.data_ptr())))
        assert ei.value.status == -4 and "static" in str(ei.value)
    finally:
        eng.close()

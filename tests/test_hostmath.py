"""Host-side helpers that need no GPU."""
import numpy as np

from image_processing_amd import hostmath


def test_flow_margin():
    m = hostmath.flow_margin([0.0, 0.1, 1.9995, 2.5, np.nan, 7.0], 2.0)
    assert abs(m["min_margin"] - 0.0005) < 1e-12 and m["frames_within_rel"] == 1
    # frame 0 (always 0: the first frame of a chunk is never discarded, ip.py:853) does not count
    assert hostmath.flow_margin([0.0], 2.0)["min_margin"] == float("inf")
    assert hostmath.flow_margin([0.0, 2.0], 2.0) == dict(min_margin=0.0, frames_within_rel=1, rel=1e-3)

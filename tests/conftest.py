import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Tests that need a second process on the GPU (multi-video replicas, the RCCL process group) fork it from a
    # helper that must exist before THIS process initialises the GPU: start it now, while nothing has.
    from image_processing_amd import multi
    multi.ensure_launcher()


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(GOLDEN, "reference_run.npz"))


def arrays_to_masks(kept, arr):
    """(kept flags, stacked bool array) -> list of bool arrays / None (ip.py:557 convention)."""
    return [arr[i].astype(bool) if kept[i] else None for i in range(len(kept))]

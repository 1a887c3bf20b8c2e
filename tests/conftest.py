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


def unpack_bits(packed, H):
    """uint8 (..., ceil(H/8)) -> bool (..., H); h=0 is the MSB of byte 0 (np.packbits default)."""
    return np.unpackbits(np.asarray(packed, dtype=np.uint8), axis=-1)[..., :H].astype(bool)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture
def golden():
    return load_golden

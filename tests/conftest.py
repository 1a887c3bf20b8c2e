import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)  # _sketch.py: the compact summaries the BASELINE-shape fixtures hold


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def unpack_bits(packed, H):
    """uint8 (..., ceil(H/8)) -> bool (..., H); h=0 is the MSB of byte 0 (np.packbits default)."""
    return np.unpackbits(np.asarray(packed, dtype=np.uint8), axis=-1)[..., :H].astype(bool)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


SHAPE_FIXTURES_SMALL = ["c2_small", "c3_small", "c4_small", "c5_small", "c3_wide"]
SHAPE_FIXTURES_FULL = ["c2", "c3", "c4", "c5"]


def sketch_close(got, want, rtol, name=""):
    """assert_allclose for _sketch.sketch values: atol relative to the largest reference entry."""
    want = np.asarray(want)
    scale = float(np.abs(want).max()) if want.size else 1.0
    np.testing.assert_allclose(np.asarray(got), want, rtol=rtol, atol=rtol * max(scale, 1e-300), err_msg=name)


@pytest.fixture
def golden():
    return load_golden

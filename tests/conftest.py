import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cvpr2023-unidistill_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("UD_RANDOM_INIT", "1")   # tests build the reference architecture on random weights
os.environ.setdefault("UD_STRICT", "1")        # a fall-through to a library convolution / GEMM raises (unidistill_amd/_lib.py)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if needed) and load the C-ABI library; CPU-safe (no kernels are launched)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-j8", "-C", PKG])
    from unidistill_amd import _lib
    return _lib.load()


@pytest.fixture
def lenient():
    """Goldens at SHRUNK channel widths (tests/golden/shrunk.py) run layers no hand-written kernel takes: library path allowed."""
    from unidistill_amd import _lib
    with _lib.strict(False):
        yield

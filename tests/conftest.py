import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The suite runs the library's DEFAULT configuration (what bench.py and users run).  The opt-in GroupNorm-backward fusion
# (diqt_conv3d_fwd_gnbwd) is switched on per test through ops.gnbwd_fuse(True) -- a runtime setter, so the whole-network gradient
# tests run in BOTH modes inside one process (tests/test_gpu_fullsize.py, tests/test_gpu_unet.py).
os.environ.pop("DIQT_GNBWD_FUSE", None)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden

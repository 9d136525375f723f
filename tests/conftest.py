import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# opt-in kernel paths that production leaves off by default run under test (read once by the library, so set before it loads):
# the GroupNorm backward's reduction in the epilogue of the conv's backward-data pass (diqt_conv3d_fwd_gnbwd)
os.environ.setdefault("DIQT_GNBWD_FUSE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden

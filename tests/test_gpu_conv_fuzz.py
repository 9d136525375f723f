"""Random small shapes through every conv_fwd9_kernel variant (DIQT_CONV_F9=2 makes it take launches of any tile count: ragged
extents in all three axes, single-tile launches, ragged channel blocks, causal temporal padding, split-K) against float64."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_random_shapes_on_the_one_wave_per_simd_conv(seed):
    env = dict(os.environ, DIQT_CONV_F9="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "f9_fuzz_worker.py"), str(seed), "36"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "FUZZ_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("kid=4") >= 20, "most cases should have run on conv_fwd9_kernel:\n" + r.stdout[-2000:]

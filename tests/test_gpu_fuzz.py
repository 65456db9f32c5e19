"""Randomised GPU parity: scenes with awkward scales (1e-3 .. 1e4), offsets up to 1e5, exact
duplicate positions, 1..1000 primitives, every bounds mode, nearest + kNN (k in 1/3/8/16) + a small
render, all against the oracle's exhaustive walk bit for bit (tools/fuzz_gpu.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_against_the_oracle(gpu, seed):
    import fuzz_gpu

    msgs = []
    bad = fuzz_gpu.run(seed, 40, gpu=gpu, log=msgs.append)
    assert bad == 0, "\n".join(msgs[:5])

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def make_points(n, extent, seed, scale=1.2):
    """n query points ~ U[-scale*extent, scale*extent)^3 with w = 1."""
    rng = np.random.default_rng(seed)
    p = np.ones((n, 4), np.float32)
    p[:, :3] = rng.uniform(-scale * extent, scale * extent, (n, 3)).astype(np.float32)
    return p


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="session")
def gpu():
    import sdf_amd
    g = sdf_amd.SdfGpu(0)
    yield g
    g.close()

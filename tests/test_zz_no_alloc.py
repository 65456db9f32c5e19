"""sdfgpu_nearest_dev_static (the allocation-free entry point) against sdfgpu_nearest_dev:
bit-identical distances and indices, two calls in a row, the second on a smaller batch (the
static scratch keeps the first call's contents)."""
import numpy as np
import pytest

import sdf_amd as S
from sdf_amd import _ffi

pytestmark = pytest.mark.gpu


def _points(n, L, seed):
    rng = np.random.default_rng(seed)
    p = np.ones((n, 4), np.float32)
    p[:, :3] = rng.uniform(-1.2 * L, 1.2 * L, (n, 3)).astype(np.float32)
    return p


def test_static_entry_point_matches_nearest_dev():
    import torch

    n_prims = 3000
    buf = S.random_scene_buffer(n_prims, 0x5DF0AA01)
    L = S.default_extent(n_prims)
    gpu = S.SdfGpu(0)
    try:
        gpu.upload_tree(buf)
        dev = torch.device("cuda", 0)
        stream = torch.cuda.current_stream().cuda_stream
        for n, seed in ((200_000, 1), (5_000, 2)):   # second call: smaller input over the same scratch
            pts = torch.from_numpy(_points(n, L, seed)).to(dev)
            d0 = torch.empty(n, dtype=torch.float32, device=dev)
            i0 = torch.empty(n, dtype=torch.int32, device=dev)
            d1 = torch.full((n,), -1.0, dtype=torch.float32, device=dev)
            i1 = torch.full((n,), -7, dtype=torch.int32, device=dev)
            gpu.nearest_dev(pts.data_ptr(), n, d0.data_ptr(), i0.data_ptr(), stream)
            gpu.nearest_dev_static(pts.data_ptr(), n, d1.data_ptr(), i1.data_ptr(), stream)
            torch.cuda.synchronize()
            assert torch.equal(d0.view(torch.int32), d1.view(torch.int32)), f"distance bits differ (n={n})"
            assert torch.equal(i0, i1), f"primitive indices differ (n={n})"
            assert int((i1 >= 0).sum()) == n
        # beyond the stated maximum: a clear error, nothing launched
        n = _ffi.STATIC_MAX_POINTS + 1
        pts = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        d1 = torch.empty(n, dtype=torch.float32, device=dev)
        with pytest.raises(S.SdfGpuError) as e:
            gpu.nearest_dev_static(pts.data_ptr(), n, d1.data_ptr(), 0, stream)
        assert e.value.status == _ffi.ERR_TOO_LARGE
    finally:
        gpu.close()

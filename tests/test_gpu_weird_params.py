"""Primitive sizes a producer should never write -- 0, negative, 1e-30, 1e30 -- patched into a valid buffer:
the arithmetic is still the reference's, so the GPU must still agree with the oracle bit for bit in the
modes that take their boxes from the primitives (BVH, AABB) or use none; NaN / infinite sizes are refused at
upload (SDFGPU_ERR_NON_FINITE).  OBB mode trusts the buffer's own boxes, which the patch leaves stale, and
is not part of this test."""
import copy

import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits

pytestmark = pytest.mark.gpu


def _patched(val, kind):
    buf = copy.deepcopy(S.random_scene_buffer(60, 0xBADF))
    dt = buf.downtree_buffer
    which = [int(i) for i in np.flatnonzero(dt["is_primitive"]) if dt[i]["op_code"] == kind][:3]
    assert which
    for i in which:
        if kind == 0:
            dt[i]["op_specific"]["floats"][0] = val      # radius (elements.rs:58-62)
        else:
            dt[i]["op_specific"]["vec4s"][0][1] = val    # one half extent (elements.rs:100-104)
    return buf


@pytest.mark.parametrize("kind", [0, 2])
@pytest.mark.parametrize("val", [0.0, -1.5, 1e-30, 1e30])
def test_degenerate_sizes_still_match_the_oracle(gpu, val, kind):
    buf = _patched(val, kind)
    rng = np.random.default_rng(0)
    pts = np.ones((4000, 4), np.float32)
    pts[:, :3] = rng.uniform(-1.2, 1.2, (4000, 3)) * S.default_extent(60)
    d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE, threads=8)
    dk0, ik0 = O.knn(buf, pts[:500], 8, threads=8)
    for bounds in (S.BOUNDS_BVH, S.BOUNDS_AABB, S.BOUNDS_NONE):
        gpu.upload_tree(buf, bounds=bounds)
        d, i = gpu.nearest(pts)
        assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
        dk, ik = gpu.knn(pts[:500], 8)
        assert np.array_equal(bits(dk), bits(dk0)) and np.array_equal(ik, ik0)


@pytest.mark.parametrize("kind", [0, 2])
@pytest.mark.parametrize("val", [float("nan"), float("inf"), float("-inf")])
def test_non_finite_sizes_are_refused(gpu, val, kind):
    buf = _patched(val, kind)
    for bounds in (S.BOUNDS_AUTO, S.BOUNDS_OBB, S.BOUNDS_NONE):
        with pytest.raises(S.SdfGpuError) as e:
            gpu.upload_tree(buf, bounds=bounds)
        assert e.value.status == -8

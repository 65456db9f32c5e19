"""Query points outside the domain (sdfgpu.h, "Domain of a query point"): a NaN or infinite coordinate,
or one beyond 2^62 in magnitude.

The reference's result for such a point is an accident of evaluation order: every comparison with a NaN
value is false, so the union's `if a < b {a} else {b}` (faux_shader.rs:27-29) keeps whatever came last, and
its prune test never fires.  The library defines the result instead -- (NaN, SDF_NO_PRIM) in every slot --
and keeps such a point out of the walk.  Checked here: that definition, on every path; that the other
points of the same batch (the same wavefronts) are bit-exact against the oracle as always; and that huge
coordinates inside the domain (1e18) still go through the ordinary path and agree with the oracle.
"""
import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits, make_points

pytestmark = pytest.mark.gpu


def _batch(n_pts, L, seed):
    pts = make_points(n_pts, L, seed=seed)
    rng = np.random.default_rng(seed)
    bad = np.sort(rng.choice(n_pts, size=max(8, n_pts // 9), replace=False))
    vals = np.array([np.nan, np.inf, -np.inf, 3e38, -1e30, 4.7e18], np.float32)   # 2^62 = 4.61e18
    for j, r in enumerate(bad):
        pts[r, j % 3] = vals[(j // 3) % len(vals)]
    pts[bad[0], :3] = np.nan
    pts[bad[1], :3] = np.inf
    huge = np.setdiff1d(np.arange(n_pts), bad)[:6]
    pts[huge[0], :3] = [4.6e18, -4.6e18, 4.6e18]
    pts[huge[1], :3] = [1e18, 0.0, 0.0]
    pts[huge[2], :3] = [0.0, -2e18, 1e18]
    return pts, bad


def _check(gpu, buf, pts, bad, k):
    ok = np.ones(len(pts), bool)
    ok[bad] = False
    d0, i0 = O.nearest(buf, pts[ok], O.MODE_EXHAUSTIVE, threads=8)
    d, i = gpu.nearest(pts)
    assert np.array_equal(bits(d[ok]), bits(d0)) and np.array_equal(i[ok], i0)
    assert np.all(np.isnan(d[bad])) and np.all(i[bad] == S.NO_PRIM)
    dk0, ik0 = O.knn(buf, pts[ok], k, threads=8)
    dk, ik = gpu.knn(pts, k)
    assert np.array_equal(bits(dk[ok]), bits(dk0)) and np.array_equal(ik[ok], ik0)
    assert np.all(np.isnan(dk[bad])) and np.all(ik[bad] == S.NO_PRIM)


@pytest.mark.parametrize("bounds", [S.BOUNDS_BVH, S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE])
@pytest.mark.parametrize("n_pts", [100, 5000])   # below / above the size at which a batch is reordered
def test_non_finite_points_are_defined_and_harmless(gpu, bounds, n_pts):
    n = 300
    buf = S.random_scene_buffer(n, 0xF1 + n)
    pts, bad = _batch(n_pts, S.default_extent(n), seed=n_pts)
    gpu.upload_tree(buf, bounds=bounds)
    _check(gpu, buf, pts, bad, 8)


def test_non_finite_points_on_the_general_path(gpu):
    buf = S.random_scene(40, 0xF2, opts={"pca_trans_inverse": True}).expanded().make_buffer()
    info = gpu.upload_tree(buf)
    assert info.identity_inner == 0
    pts, bad = _batch(700, S.default_extent(40), seed=3)
    _check(gpu, buf, pts, bad, 3)


def test_a_batch_of_nothing_but_nan_costs_nothing(gpu):
    # no lane drives a walk: the call must come back at once, not after an exhaustive walk per wavefront
    buf = S.random_scene_buffer(2000, 0xF3)
    gpu.upload_tree(buf)
    pts = np.full((200_000, 4), np.nan, np.float32)
    pts[:, 3] = 1.0
    d, i = gpu.nearest(pts)
    assert np.all(np.isnan(d)) and np.all(i == S.NO_PRIM)
    dk, ik = gpu.knn(pts, 8)
    assert np.all(np.isnan(dk)) and np.all(ik == S.NO_PRIM)

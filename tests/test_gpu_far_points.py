"""Points far outside the scene (beyond the range of the scaled, clamped box test of the near walk).

Inside ~15 scene extents a per-axis box distance is one subtraction with the hardware's [0,1] result clamp;
further out every box would look the same sqrt(3) scaled units away, nothing would be pruned and a
wavefront would visit the whole tree.  A walk with such a driving lane runs the unclamped copy instead
(kernels.hip, aabb_dist2 / FAR).  Checked: results bit-exact against the oracle at 20 ... 1e6 extents, for
both walks (assembly, C++ state machine) and through the march (a camera far away); and the visit counts,
which must stay those of a pruned walk.
"""
import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits, make_points

pytestmark = pytest.mark.gpu


def _far_points(n_pts, L, factor, seed):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n_pts, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.ones((n_pts, 4), np.float32)
    pts[:, :3] = (d * L * factor * rng.uniform(0.8, 1.2, (n_pts, 1))).astype(np.float32)
    return pts


@pytest.mark.parametrize("bounds", [S.BOUNDS_BVH, S.BOUNDS_AABB])
@pytest.mark.parametrize("factor", [20.0, 300.0, 1e6])
def test_far_points_match_the_oracle(gpu, bounds, factor):
    n = 1000
    L = S.default_extent(n)
    buf = S.random_scene_buffer(n, 0xFA0 + n)
    gpu.upload_tree(buf, bounds=bounds)
    pts = _far_points(6000, L, factor, seed=int(factor))
    pts[::7] = make_points(len(pts[::7]), L, seed=3)          # near points in the same wavefronts
    d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE, threads=8)
    d, i = gpu.nearest(pts)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    dk0, ik0 = O.knn(buf, pts[:1500], 8, threads=8)
    dk, ik = gpu.knn(pts[:1500], 8)
    assert np.array_equal(bits(dk), bits(dk0)) and np.array_equal(ik, ik0)


@pytest.mark.parametrize("asm", [0, 1])
def test_far_points_are_still_pruned(gpu, asm):
    import torch
    from sdf_amd import _ffi

    n = 5000
    L = S.default_extent(n)
    buf = S.random_scene_buffer(n, 0xFA1)
    gpu.upload_tree(buf)
    near = torch.from_numpy(make_points(20_000, L, seed=1)).cuda()
    far = torch.from_numpy(_far_points(20_000, L, 40.0, seed=2)).cuda()
    try:
        gpu.set_option(_ffi.OPT_STATS_ASM_WALK, asm)
        a = gpu.nearest_stats_dev(near.data_ptr(), near.shape[0])
        b = gpu.nearest_stats_dev(far.data_ptr(), far.shape[0])
    finally:
        gpu.set_option(_ffi.OPT_STATS_ASM_WALK, 0)
    waves = (20_000 + 63) // 64
    per_wave_near = a["wave_inner"] / waves
    per_wave_far = b["wave_inner"] / waves
    # an unpruned walk would visit all n - 1 internal nodes for every wavefront
    assert per_wave_far < 0.2 * (n - 1), (per_wave_near, per_wave_far)


def test_camera_far_from_the_scene(gpu):
    n = 300
    L = S.default_extent(n)
    buf = S.random_scene_buffer(n, 0xFA2)
    w, h = 96, 64
    cam = S.camera_look_at((0.0, 0.0, -60.0 * L), (0.0, 0.0, 0.0), (0, 1, 0), 0.03, w, h)
    mp = S.march_params(max_steps=48, eps_hit=1e-3, t_min=0.0, t_max=200.0 * L, normal_h=1e-3)
    ref = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    assert ref["hit"].sum() > 100
    for bounds in (S.BOUNDS_BVH, S.BOUNDS_AABB):
        gpu.upload_tree(buf, bounds=bounds)
        img = gpu.render(cam, w, h, mp)
        assert np.array_equal(img["hit"], ref["hit"]) and np.array_equal(img["prim"], ref["prim"])
        assert np.array_equal(bits(img["depth"]), bits(ref["depth"]))
        assert np.array_equal(bits(img["normal"]), bits(ref["normal"]))

"""GPU parity: sdfgpu_nearest (K1) against the CPU restatement of
faux_shader::nearest_neighbor (sdf/src/faux_shader.rs:96-208).

Bar: distances bit-exact (stronger than the 1e-5 relative the north star
asks), primitive indices bit-exact.  All calls go through the C ABI.
"""
import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits, make_points

pytestmark = pytest.mark.gpu


def _check(gpu, buf, pts, bounds, mode=O.MODE_EXHAUSTIVE):
    gpu.upload_tree(buf, bounds=bounds)
    d, i = gpu.nearest(pts)
    d0, i0 = O.nearest(buf, pts, mode)
    assert np.array_equal(bits(d), bits(d0)), f"distance bits differ at {np.flatnonzero(bits(d) != bits(d0))[:8]}"
    assert np.array_equal(i, i0), f"indices differ at {np.flatnonzero(i != i0)[:8]}"


@pytest.mark.parametrize("bounds", [S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE, S.BOUNDS_BVH])
@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 257, 1000])
def test_nearest_matches_oracle(gpu, n, bounds):
    buf = S.random_scene_buffer(n, 0x5DF00000 + n)
    L = S.default_extent(n)
    pts = make_points(4096 if n < 1000 else 16384, L, seed=n)
    _check(gpu, buf, pts, bounds)


def test_nearest_matches_faux_as_written(gpu):
    # the literal walk (with the reference's own leaf-pair prune) agrees with
    # the exhaustive one on consistent buffers, and so does the GPU
    buf = S.random_scene_buffer(300, 77)
    pts = make_points(4096, S.default_extent(300), seed=5)
    _check(gpu, buf, pts, S.BOUNDS_AUTO, mode=O.MODE_FAUX)


def test_config1_64k_points_1k_prims(gpu):
    # BASELINE config 1: 1k primitives, seed 0x5DF00001, 64 Ki points
    buf = S.random_scene_buffer(1000, 0x5DF00001)
    pts = make_points(65536, S.default_extent(1000), seed=0x5DF00001)
    gpu.upload_tree(buf)
    d, i = gpu.nearest(pts)
    d0, i0 = O.nearest(buf, pts, O.MODE_TUNED, threads=8)
    assert np.array_equal(bits(d), bits(d0))
    assert np.array_equal(i, i0)


def test_empty_tree(gpu):
    gpu.upload_tree(S.SdfTreeBuffer.make_empty())
    d, i = gpu.nearest(make_points(100, 10.0, 1))
    assert np.all(np.isposinf(d)) and np.all(i == S.NO_PRIM)   # faux_shader.rs:104,207


def test_ragged_point_counts(gpu):
    buf = S.random_scene_buffer(50, 9)
    for n in (1, 63, 64, 65, 255, 256, 257, 1000):
        _check(gpu, buf, make_points(n, S.default_extent(50), seed=n), S.BOUNDS_AUTO)


def test_points_inside_primitives_and_far_away(gpu):
    buf = S.random_scene_buffer(200, 31)
    L = S.default_extent(200)
    near = make_points(4096, L, seed=3, scale=0.3)      # many inside / close
    far = make_points(4096, L, seed=4, scale=40.0)      # far outside the scene
    # beyond 16 scene extents the box test's [0,1] result clamp saturates (kernels.hip, aabb_dist2):
    # the bound only gets looser there, results must not change
    very_far = make_points(1024, L, seed=5, scale=3000.0)
    huge = make_points(256, L, seed=6, scale=1e12)
    for b in (S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_BVH):
        _check(gpu, buf, near, b)
        _check(gpu, buf, far, b)
        _check(gpu, buf, very_far, b)
        _check(gpu, buf, huge, b)


def test_duplicate_primitive_tie_goes_to_later(gpu):
    # two identical spheres: the reference's min is `if a < b {a} else {b}`
    # (faux_shader.rs:27-29) so the later one in pre-order wins
    t = S.Transform.from_translation((1.0, 2.0, 3.0))
    b = (S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t).operation(S.SdfUnion(0.0))
         .with_(S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t))
         .with_(S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(S.Transform.from_translation((30.0, 0, 0)))))
    buf = b.finalize().expanded().make_buffer()
    pts = make_points(2048, 20.0, seed=8)
    _check(gpu, buf, pts, S.BOUNDS_AABB, mode=O.MODE_FAUX)
    _check(gpu, buf, pts, S.BOUNDS_OBB, mode=O.MODE_EXHAUSTIVE)
    gpu.upload_tree(buf)
    d, i = gpu.nearest(pts)
    dt = buf.downtree_buffer
    # the two coincident spheres are two of the three primitives; find them
    prim_nodes = np.flatnonzero(dt["is_primitive"])
    tr = dt["bounding_box"]["trans_inverse"][prim_nodes][:, 12:15]
    same = [k for k in range(3) if np.allclose(tr[k], [-1, -2, -3])]
    assert len(same) == 2
    near_dup = np.isin(i, same)
    assert near_dup.any() and np.all(i[near_dup] == max(same))


def test_points_exactly_on_centres_and_tiny_offsets(gpu):
    # squared lengths of exactly 0 and far below 2^-96: the wavefronts holding such points take the
    # compiler's full square-root expansion, the others the short one (kernels.hip, sqrt_rn) -- a batch
    # mixes both, and a render marches through the same code
    import itertools
    b = S.SdfBuilder.primitive(S.SdfSphere(0.75)).transform(S.Transform.from_translation((0.0, 0.0, 0.0))).operation(S.SdfUnion(0.0))
    centres = [(0.0, 0.0, 0.0)]
    for c in itertools.product((-8.0, 8.0), (-4.0, 4.0), (-2.0, 2.0)):
        centres.append(c)
        b = b.with_(S.SdfBuilder.primitive(S.SdfSphere(0.5) if c[0] < 0 else S.SdfBox((0.5, 0.25, 1.0)))
                    .transform(S.Transform.from_translation(c)))
    buf = b.finalize().expanded().make_buffer()
    pts = make_points(4096, 10.0, seed=21)
    c = np.array(centres, np.float32)
    pts[:9, :3] = c                                             # squared length exactly 0
    pts[64:73, :3] = c + np.float32(1e-20)                      # normal-range coordinates, squared length denormal / 0
    pts[128:137, :3] = c * np.float32(1.0) + np.array([1e-30, 0, 0], np.float32)
    pts[200:209, :3] = c + np.array([0.0, 1e-15, 0.0], np.float32)   # squared length 1e-30 < 2^-96
    for bounds in (S.BOUNDS_BVH, S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE):
        _check(gpu, buf, pts, bounds)


def test_w_not_one_is_rejected(gpu):
    gpu.upload_tree(S.random_scene_buffer(4, 1))
    p = make_points(8, 10.0, 1)
    p[3, 3] = 0.5
    with pytest.raises(S.SdfGpuError) as e:
        gpu.nearest(p)
    assert e.value.status == S._ffi.ERR_POINT_W


@pytest.mark.parametrize("chain", [False, True])
def test_dense_nn_single_and_chain_on_the_gpu(gpu, chain):
    # the crate's test_dense_nn_{single,chain}_{uni,dir} (node.rs:625-656) through the GPU path:
    # one primitive under translate / rotate / both orders, bare or inside a one-slot union
    import math
    rng = np.random.default_rng(9)
    for prim in (S.SdfSphere(1.0), S.SdfBox((0.7, 1.3, 2.1))):
        for _ in range(6):
            tl = S.Transform.from_translation(rng.uniform(-50, 50, 3))
            rot = S.Transform.from_rotation(S.Quat.from_euler_xyz(*rng.uniform(0, 2 * math.pi, 3)))
            TL, RO = np.linalg.inv(tl.compute_matrix()), np.linalg.inv(rot.compute_matrix())
            pts = np.ones((64, 4), np.float32)
            pts[:, :3] = rng.uniform(-50, 50, (64, 3))
            for ts, inv in (([tl], TL), ([rot], RO), ([tl, rot], TL @ RO), ([rot, tl], RO @ TL)):
                b = S.SdfBuilder.primitive(prim)
                for t in ts:
                    b = b.transform(t)
                if chain:
                    b = b.operation(S.SdfUnion(0.0))
                buf = b.finalize().expanded().make_buffer()
                gpu.upload_tree(buf)
                d, i = gpu.nearest(pts)
                d0, i0 = O.nearest(buf, pts, mode=O.MODE_FAUX)
                assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
                gt = np.array([prim.distance_to((inv @ q.astype(np.float64))[:3]) for q in pts])
                assert np.all(np.abs(d - gt) <= 4e-5 * np.maximum(1.0, np.abs(gt)))


def test_buffer_deeper_than_the_wave_stack(gpu):
    # 100 nested unions: deeper than the 64-entry wave stack, fine for the reference's 256 frames
    from test_compact_validation import deep_chain
    buf = deep_chain(100)
    pts = make_points(4096, 150.0, seed=9)
    pts[:, 0] = np.abs(pts[:, 0]) * 1.7
    _check(gpu, buf, pts, S.BOUNDS_AUTO)


@pytest.mark.parametrize("n,reorder", [(300, True), (5000, True), (5000, False)])
def test_assembly_walk_visits_what_the_cxx_walk_visits(gpu, n, reorder):
    # the production kernels run the hand-scheduled walk, the roofline units are counted by the C++
    # state machine: both must visit the same internal nodes and leaves, wavefront by wavefront
    import torch
    from sdf_amd import _ffi

    buf = S.random_scene_buffer(n, 0xA5A5 + n)
    gpu.upload_tree(buf)
    gpu.set_reorder(reorder)
    pts = torch.from_numpy(make_points(50_000, S.default_extent(n), seed=n)).cuda()
    try:
        gpu.set_option(_ffi.OPT_STATS_ASM_WALK, 0)
        a = gpu.nearest_stats_dev(pts.data_ptr(), pts.shape[0])
        gpu.set_option(_ffi.OPT_STATS_ASM_WALK, 1)
        b = gpu.nearest_stats_dev(pts.data_ptr(), pts.shape[0])
    finally:
        gpu.set_option(_ffi.OPT_STATS_ASM_WALK, 0)
        gpu.set_reorder(True)
    assert a["wave_inner"] > 0 and a["wave_leaf"] > 0
    for key in ("queries", "wave_inner", "wave_leaf"):
        assert a[key] == b[key], (key, a[key], b[key])

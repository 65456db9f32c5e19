"""GPU parity on the general path (non-identity internal trans_inverse, projective
rows: SURVEY C-3) and of the reference's bounds MINDIST / MAXDIST (A-2, A-3)."""
import math

import numpy as np
import pytest

import sdf_amd as S
from sdf_amd import _ffi
from oracle import oracle as O
from conftest import bits, make_points

pytestmark = pytest.mark.gpu


def pca_scene(n, seed):
    return S.random_scene(n, seed, opts={"pca_trans_inverse": True}).expanded().make_buffer()


@pytest.mark.parametrize("n", [2, 3, 9, 64, 300])
def test_general_nearest_and_knn(gpu, n):
    buf = pca_scene(n, 0x6E00 + n)
    pts = make_points(2000, S.default_extent(n), n)
    d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE, threads=8)
    info = gpu.upload_tree(buf, bounds=S.BOUNDS_NONE)      # the exhaustive general path
    assert info.identity_inner == 0 and info.bounds == S.BOUNDS_NONE
    d, i = gpu.nearest(pts)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    info = gpu.upload_tree(buf)                             # AUTO: the pruned chain path (round 3)
    assert info.identity_inner == 0 and info.bounds == S.BOUNDS_BVH
    d, i = gpu.nearest(pts)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    # sdfgpu.h promises "the value the reference returns": on these buffers the as-written walk, whose
    # union prune (faux_shader.rs:117-126) tests boxes in a frame the re-based point has left, must
    # agree with the exhaustive one -- an empirical property, so it is checked, not assumed
    df, fi = O.nearest(buf, pts, O.MODE_FAUX, threads=8)
    assert np.array_equal(bits(df), bits(d0)) and np.array_equal(fi, i0)
    ff, _ = O.nearest(buf, pts[:64], O.MODE_FAUX_FAITHFUL)
    assert np.array_equal(bits(ff), bits(d0[:64]))
    for k in (3, 8):
        kd, ki = gpu.knn(pts[:500], k)
        kd0, ki0 = O.knn(buf, pts[:500], k, threads=8)
        assert np.array_equal(bits(kd), bits(kd0)) and np.array_equal(ki, ki0)


def test_general_differs_from_identity_scene(gpu):
    # the re-basing really is applied: same primitives, different answers than the identity build
    a = S.random_scene_buffer(20, 5)
    b = pca_scene(20, 5)
    pts = make_points(500, S.default_extent(20), 1)
    gpu.upload_tree(a)
    da, _ = gpu.nearest(pts)
    gpu.upload_tree(b)
    db, _ = gpu.nearest(pts)
    assert not np.array_equal(bits(da), bits(db))


def test_projective_leaf_row(gpu):
    # bottom-right element 1 + ulp: w is carried through the chain like the reference does
    buf = S.random_scene_buffer(12, 8)
    dt = buf.downtree_buffer.copy()
    leaf = int(np.flatnonzero(dt["is_primitive"])[3])
    dt["bounding_box"]["trans_inverse"][leaf][15] = np.nextafter(np.float32(1), np.float32(2))
    inner = int(np.flatnonzero(dt["is_primitive"] == 0)[1])
    dt["bounding_box"]["trans_inverse"][inner][15] = np.float32(0.75)     # children see w = 0.75
    buf2 = S.SdfTreeBuffer(dt, buf.uptree_buffer)
    info = gpu.upload_tree(buf2)
    assert info.bounds == S.BOUNDS_NONE
    pts = make_points(1500, S.default_extent(12), 2)
    d, i = gpu.nearest(pts)
    d0, i0 = O.nearest(buf2, pts, O.MODE_EXHAUSTIVE)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)


def test_general_render(gpu):
    n, w, h = 25, 48, 32
    buf = pca_scene(n, 77)
    L = S.default_extent(n)
    cam = S.camera_look_at((0, 0, -3.5 * L), (0, 0, 0), (0, 1, 0), math.pi / 3, w, h)
    mp = S.march_params(max_steps=96, eps_hit=1e-3, t_min=0.0, t_max=10 * L, normal_h=1e-3)
    ref = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    assert ref["hit"].sum() > 0
    for bounds in (S.BOUNDS_AUTO, S.BOUNDS_NONE):           # the pruned chain path, the exhaustive general path
        gpu.upload_tree(buf, bounds=bounds)
        img = gpu.render(cam, w, h, mp)
        assert np.array_equal(img["hit"], ref["hit"]) and np.array_equal(img["prim"], ref["prim"])
        assert np.array_equal(bits(img["depth"]), bits(ref["depth"]))
        assert np.array_equal(bits(img["normal"]), bits(ref["normal"]))


@pytest.mark.parametrize("n,seed", [(700, 11), (3000, 12)])
def test_chain_path_at_size(gpu, n, seed):
    """Crate-style buffers (PCA trans_inverse in every merged box, obb.rs:183-185) large enough for a deep hierarchy,
    the candidate grid and many-level chains: nearest, kNN and a frame, bit for bit against the exhaustive oracle and
    against the library's own exhaustive general path."""
    buf = pca_scene(n, seed)
    L = S.default_extent(n)
    pts = make_points(6000, L, seed)
    d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE, threads=8)
    info = gpu.upload_tree(buf)
    assert info.bounds == S.BOUNDS_BVH and info.identity_inner == 0 and info.grid_cells > 0
    d, i = gpu.nearest(pts)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    kd, ki = gpu.knn(pts[:2000], 8)
    kd0, ki0 = O.knn(buf, pts[:2000], 8, threads=8)
    assert np.array_equal(bits(kd), bits(kd0)) and np.array_equal(ki, ki0)
    w, h = 160, 96
    cam = S.camera_look_at((0.2 * L, 0.1 * L, -2.5 * L), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
    mp = S.march_params(max_steps=128, eps_hit=1e-3, t_min=0.0, t_max=6 * L, normal_h=1e-4)
    img = gpu.render(cam, w, h, mp)
    gpu.upload_tree(buf, bounds=S.BOUNDS_NONE)
    ref = gpu.render(cam, w, h, mp)                         # the exhaustive general path (itself pinned to the oracle above)
    assert ref["hit"].sum() > 50
    for key in ("hit", "prim", "depth", "normal"):
        assert np.array_equal(np.ascontiguousarray(img[key]).view(np.uint8), np.ascontiguousarray(ref[key]).view(np.uint8)), key
    sub = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8, pix_first=0, pix_count=w * h // 7, pix_stride=7)
    sel = np.arange(0, w * h, 7)[: w * h // 7]
    assert np.array_equal(img["hit"][sel], sub["hit"]) and np.array_equal(bits(img["depth"][sel]), bits(sub["depth"]))
    assert np.array_equal(img["prim"][sel], sub["prim"]) and np.array_equal(bits(img["normal"][sel]), bits(sub["normal"]))


def test_box_bounds_match_the_reference_formulas(gpu):
    # A-2 mindist (faux_shader.rs:54-58) and A-3 maxdist (:39-52), bit for bit
    buf = S.random_scene_buffer(60, 3)
    pts = make_points(4096, S.default_extent(60), 4, scale=1.5)
    for node in (0, 1, 2, 17, 60, 118):
        lo, hi = gpu.box_bounds(buf.downtree_buffer["bounding_box"][node], pts)
        lo0, hi0 = O.bounds(buf, node, pts)
        assert np.array_equal(bits(lo), bits(lo0)) and np.array_equal(bits(hi), bits(hi0))
    # a box with a projective / scaled w lane exercises all four lanes
    blk = buf.downtree_buffer["bounding_box"][5].copy()
    blk["scale"][3] = 0.5
    blk["full_inverse"][15] = 1.25
    odd = S.SdfTreeBuffer(buf.downtree_buffer.copy(), buf.uptree_buffer)
    odd.downtree_buffer["bounding_box"][5] = blk
    lo, hi = gpu.box_bounds(blk, pts)
    lo0, hi0 = O.bounds(odd, 5, pts)
    assert np.array_equal(bits(lo), bits(lo0)) and np.array_equal(bits(hi), bits(hi0))


# ---- clone op (faux_shader.rs:75-78,90): a clone subtree is +inf as written ----

def _clone_mix(n_live=40, n_clones=6, seed=21):
    rng = np.random.default_rng(seed)
    L = 25.0

    def prim():
        e = S.SdfSphere(float(rng.uniform(0.5, 2.0))) if rng.random() < 0.5 else S.SdfBox(tuple(float(v) for v in rng.uniform(0.5, 2.0, 3)))
        return S.SdfBuilder.primitive(e).transform(S.Transform.from_translation(tuple(float(v) for v in rng.uniform(-L, L, 3))))

    parts = [prim() for _ in range(n_live)]
    for _ in range(n_clones):
        parts.insert(int(rng.integers(0, len(parts))),
                     prim().operation(S.SdfCaaClone((6, 6, 6), (-2, -2, -2), (2, 2, 2))))
    b = parts[0].operation(S.SdfUnion(0.0))
    for x in parts[1:]:
        b = b.with_(x)
    return b.finalize().expanded().make_buffer(), L


@pytest.mark.parametrize("bounds", [S.BOUNDS_AUTO, S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE, S.BOUNDS_BVH])
def test_clone_subtrees_are_infinite_on_the_gpu_too(gpu, bounds):
    from conftest import make_points
    buf, L = _clone_mix()
    pts = make_points(4000, L, 4)
    d0, i0 = O.nearest(buf, pts, mode=O.MODE_EXHAUSTIVE, threads=8)
    gpu.upload_tree(buf, bounds=bounds)
    d, i = gpu.nearest(pts)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    dk, ik = gpu.knn(pts[:1500], 8)
    dk0, ik0 = O.knn(buf, pts[:1500], 8, threads=8)
    assert np.array_equal(bits(dk), bits(dk0)) and np.array_equal(ik, ik0)
    w, h = 96, 64
    cam = S.camera_look_at((0.3 * L, 0.2 * L, -2.5 * L), (0, 0, 0), (0, 1, 0), 0.8, w, h)
    mp = S.march_params(max_steps=96, eps_hit=1e-3, t_min=0.0, t_max=8 * L, normal_h=1e-4)
    img = gpu.render(cam, w, h, mp)
    ref = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    assert ref["hit"].sum() > 0
    for k in ("hit", "prim"):
        assert np.array_equal(img[k], ref[k])
    assert np.array_equal(bits(img["depth"]), bits(ref["depth"])) and np.array_equal(bits(img["normal"]), bits(ref["normal"]))


def test_clone_only_tree_on_the_gpu(gpu):
    from conftest import make_points
    node = S.SdfBuilder.primitive(S.SdfSphere(2.0)).operation(S.SdfCaaClone((5, 5, 5), (0, 0, 0), (3, 3, 3))).finalize()
    buf = node.expanded().make_buffer()
    gpu.upload_tree(buf)
    d, i = gpu.nearest(make_points(3000, 10.0, 2))
    assert np.all(np.isinf(d)) and np.all(i == S.NO_PRIM)
    cam = S.camera_look_at((0, 0, -20.0), (0, 0, 0), (0, 1, 0), 0.8, 32, 24)
    mp = S.march_params(max_steps=16, eps_hit=1e-3, t_min=0.0, t_max=100.0, normal_h=1e-4)
    assert gpu.render(cam, 32, 24, mp)["hit"].sum() == 0

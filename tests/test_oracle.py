"""The oracle (CPU restatement of sdf/src/faux_shader.rs) against the
hand-derived known answers of SURVEY Appendix E, against an independent
float64 numpy evaluation, and against itself (as-written == exhaustive ==
tuned on consistent buffers).  The reference holds no golden vectors and
cannot be built here, so these are the pins there are (parity unpinned)."""
import math

import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits, make_points


def col_major(m16):
    return np.asarray(m16, np.float64).reshape(4, 4).T


def numpy_reference(buf, pts):
    """Independent float64 evaluation: min over primitives of prim(T^-1 p)."""
    dt = buf.downtree_buffer
    prims = np.flatnonzero(dt["is_primitive"])
    P = pts.astype(np.float64)
    best = np.full(len(P), np.inf)
    arg = np.zeros(len(P), np.int64)
    for k, i in enumerate(prims):
        T = col_major(dt["bounding_box"]["trans_inverse"][i])
        q = (T @ P.T).T[:, :3]
        if dt["op_code"][i] == 0:
            d = np.linalg.norm(q, axis=1) - float(dt["op_specific"]["floats"][i][0])
        else:
            h = dt["op_specific"]["vec4s"][i][0][:3].astype(np.float64)
            a = np.abs(q) - h
            d = np.linalg.norm(np.maximum(a, 0), axis=1) + np.minimum(a.max(axis=1), 0)
        upd = d < best
        best[upd], arg[upd] = d[upd], k
    return best, arg


def test_single_translated_sphere_E1():
    t = np.array([3.0, -4.0, 5.5])
    buf = (S.SdfBuilder.primitive(S.SdfSphere(1.0)).transform(S.Transform.from_translation(t))
           .finalize().expanded().make_buffer())
    pts = make_points(256, 50.0, 1, scale=1.0)
    d, i = O.nearest(buf, pts)
    want = np.linalg.norm(pts[:, :3].astype(np.float64) - t, axis=1) - 1.0
    assert np.allclose(d, want, rtol=0, atol=2e-5) and np.all(i == 0)


def test_empty_buffer_E6():
    d, i = O.nearest(S.SdfTreeBuffer.make_empty(), make_points(5, 1.0, 1))
    assert np.all(np.isposinf(d)) and np.all(i == S.NO_PRIM)


def test_dense_nn_single_uni_and_dir():
    """Mirror of the crate's test_dense_nn_single_{uni,dir} (node.rs:506-580, 625-640):
    translate / rotate / translate->rotate / rotate->translate of one primitive;
    the tree result equals prim.distance_to(M^-1 p).  The box stands in for the
    crate's direction-dependent TestPrimitive."""
    rng = np.random.default_rng(2024)
    for prim in (S.SdfSphere(1.0), S.SdfBox((0.7, 1.3, 2.1))):
        for _ in range(20):
            tl = S.Transform.from_translation(rng.uniform(-50, 50, 3))
            rot = S.Transform.from_rotation(S.Quat.from_euler_xyz(*rng.uniform(0, 2 * math.pi, 3)))
            p = np.ones((1, 4), np.float32)
            p[0, :3] = rng.uniform(-50, 50, 3)
            TL, RO = np.linalg.inv(tl.compute_matrix()), np.linalg.inv(rot.compute_matrix())
            cases = {
                "tlate": ([tl], TL), "rot": ([rot], RO),
                "tlate_rot": ([tl, rot], TL @ RO), "rot_tlate": ([rot, tl], RO @ TL),   # node.rs:536-537
            }
            for name, (chain, inv) in cases.items():
                b = S.SdfBuilder.primitive(prim)
                for t in chain:
                    b = b.transform(t)
                buf = b.finalize().expanded().make_buffer()
                gt = prim.distance_to((inv @ p[0].astype(np.float64))[:3])
                got = O.nearest(buf, p)[0][0]
                assert abs(got - gt) <= 4e-5 * max(1.0, abs(gt)), (name, got, gt)


def test_two_leaf_union_E4_and_prune():
    a = S.Transform.from_translation((-5.0, 0, 0))
    b = S.Transform.from_translation((5.0, 0, 0))
    buf = (S.SdfBuilder.primitive(S.SdfSphere(1.0)).transform(a).operation(S.SdfUnion(0.0))
           .with_(S.SdfBuilder.primitive(S.SdfSphere(2.0)).transform(b)).finalize().expanded().make_buffer())
    dt = buf.downtree_buffer
    assert list(dt["level"]) == [1, 2, 2] and list(dt["len"]) == [2, 0, 0]
    assert list(buf.uptree_buffer["level"]) == [2, 2, 1]
    centres = -dt["bounding_box"]["trans_inverse"][1:, 12:15].astype(np.float64)
    radii = dt["op_specific"]["floats"][1:, 0].astype(np.float64)
    pts = make_points(512, 10.0, 3)
    d, i = O.nearest(buf, pts)
    dd = np.linalg.norm(pts[:, None, :3] - centres[None], axis=2) - radii[None]
    assert np.allclose(d, dd.min(axis=1), atol=1e-5)
    clear = np.abs(dd[:, 0] - dd[:, 1]) > 1e-4
    assert np.array_equal(i[clear], dd.argmin(axis=1)[clear])


@pytest.mark.parametrize("n,seed", [(3, 1), (7, 2), (64, 3), (500, 4)])
def test_as_written_equals_exhaustive_equals_tuned(n, seed):
    buf = S.random_scene_buffer(n, seed)
    pts = make_points(3000, S.default_extent(n), seed)
    d0, i0 = O.nearest(buf, pts, O.MODE_FAUX)
    d1, i1 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
    d2, i2 = O.nearest(buf, pts, O.MODE_TUNED, threads=4)
    d3, i3 = O.nearest(buf, pts[:200], O.MODE_FAUX_FAITHFUL)
    assert np.array_equal(bits(d0), bits(d1)) and np.array_equal(i0, i1)
    assert np.array_equal(bits(d2), bits(d1)) and np.array_equal(i2, i1)
    assert np.array_equal(bits(d3), bits(d1[:200])) and np.array_equal(i3, i1[:200])


def test_against_float64_numpy():
    n = 300
    buf = S.random_scene_buffer(n, 11)
    pts = make_points(4000, S.default_extent(n), 12)
    d, i = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
    ref, arg = numpy_reference(buf, pts)
    assert np.allclose(d, ref, rtol=1e-5, atol=2e-4)
    second = np.partition(np.stack([numpy_reference(buf, pts)[0]]), 0)[0]
    agree = i == arg
    assert agree.mean() > 0.999          # disagreements only on near-ties
    assert np.all(np.abs(d[~agree] - ref[~agree]) < 2e-4)


def test_tie_goes_to_the_later_primitive():
    t = S.Transform.from_translation((1.0, 2.0, 3.0))
    buf = (S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t).operation(S.SdfUnion(0.0))
           .with_(S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t)).finalize().expanded().make_buffer())
    d, i = O.nearest(buf, make_points(64, 10.0, 1), O.MODE_EXHAUSTIVE)
    assert np.all(i == 1)                 # faux_shader.rs:27-29: `if a < b {a} else {b}`
    d2, i2 = O.nearest(buf, make_points(64, 10.0, 1), O.MODE_FAUX)
    assert np.array_equal(i2, i) and np.array_equal(bits(d2), bits(d))


def test_bounds_follow_the_reference_formulas():
    # MINDIST (faux_shader.rs:54-58) / MAXDIST (:39-52) against the host builder's
    # SdfBoundingBox::distance_to / max_distance (obb.rs:294-307) and analytic values
    buf = S.SdfBuilder.primitive(S.SdfSphere(2.0)).finalize().expanded().make_buffer()
    p = np.array([[5, 0, 0, 1], [0, 0, 0, 1], [3, 3, 3, 1]], np.float32)
    lo, hi = O.bounds(buf, 0, p)
    assert np.allclose(lo, [3.0, -2.0, math.sqrt(3.0)], atol=1e-6)
    assert np.allclose(hi, [np.linalg.norm([7, 2, 2]), np.linalg.norm([2, 2, 2]), np.linalg.norm([5, 5, 5])], atol=1e-5)
    from sdf_amd import _ffi
    lib = _ffi.load_library()
    big = S.random_scene_buffer(40, 8)
    pts = make_points(64, 60.0, 9)
    for node in (0, 1, 5, 17):
        lo, hi = O.bounds(big, node, pts)
        blk = np.array([big.downtree_buffer["bounding_box"][node]])
        for k in range(len(pts)):
            q = np.ascontiguousarray(pts[k, :3])
            assert abs(lib.sdfhost_bbox_distance_to(blk.ctypes.data, _ffi.fptr(q)) - lo[k]) <= 1e-5 * max(1, abs(lo[k]))
            assert abs(lib.sdfhost_bbox_max_distance(blk.ctypes.data, _ffi.fptr(q)) - hi[k]) <= 1e-5 * max(1, abs(hi[k]))


def test_knn_oracle_matches_numpy_sort():
    n, k = 120, 8
    buf = S.random_scene_buffer(n, 21)
    pts = make_points(200, S.default_extent(n), 22)
    d, i = O.knn(buf, pts, k)
    assert d.shape == (200, k) and np.all(np.diff(d, axis=1) >= 0)
    d1, i1 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
    assert np.array_equal(bits(d[:, 0]), bits(d1)) and np.array_equal(i[:, 0], i1)
    for q in range(0, 200, 17):
        all_d = np.array([O.eval_prim(buf, j, pts[q:q + 1])[0] for j in range(n)])
        order = sorted(range(n), key=lambda j: (all_d[j], -j))[:k]
        assert list(i[q]) == order
    # fewer primitives than k: padded with (+inf, NO_PRIM)
    small = S.random_scene_buffer(3, 5)
    d, i = O.knn(small, pts[:4], k)
    assert np.all(np.isposinf(d[:, 3:])) and np.all(i[:, 3:] == S.NO_PRIM)


def test_render_modes_agree_and_normals_are_unit():
    n, w, h = 30, 48, 32
    buf = S.random_scene_buffer(n, 77)
    L = S.default_extent(n)
    cam = S.camera_look_at((0, 0, -2.5 * L), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
    mp = S.march_params(128, 1e-3, 0.0, 6 * L, 1e-2)
    a = O.render(buf, cam, w, h, mp, mode=O.MODE_FAUX, threads=4)
    b = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=4)
    c = O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, threads=4)
    for k in ("hit", "prim", "depth", "normal", "steps"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
        assert np.array_equal(c[k].view(np.uint8), b[k].view(np.uint8)), k
    hit = a["hit"].astype(bool)
    assert 0 < hit.sum() < w * h
    assert np.allclose(np.linalg.norm(a["normal"][hit], axis=1), 1.0, atol=1e-5)
    assert np.all(a["normal"][~hit] == 0) and np.all(np.isposinf(a["depth"][~hit])) and np.all(a["prim"][~hit] == S.NO_PRIM)
    # strided sampling renders the same pixels
    s = O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, pix_first=3, pix_stride=7, pix_count=50)
    assert np.array_equal(s["depth"].view(np.uint32), a["depth"][3:3 + 7 * 50:7].view(np.uint32))


def test_tuned_knn_equals_exhaustive_knn():
    # the pruned k-nearest walk (the kNN leg's CPU baseline) returns what the exhaustive top-k returns
    for n, seed in ((1, 5), (5, 6), (300, 7)):
        buf = S.random_scene_buffer(n, seed)
        pts = make_points(500, S.default_extent(n), seed)
        for k in (1, 3, 8):
            d0, i0 = O.knn(buf, pts, k)
            d1, i1 = O.knn(buf, pts, k, tuned=True, threads=2)
            assert np.array_equal(bits(d0), bits(d1)) and np.array_equal(i0, i1)

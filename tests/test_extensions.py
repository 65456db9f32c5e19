"""The opt-in extensions of SURVEY 8f N-2 (include/sdfgpu.h sdfgpu_extension): smooth union, the box
frame primitive (op 5) and the repaired clone.  The crate carries the data of all three and evaluates
none (faux_shader.rs:65,87,90), so they are build-defined: default = as written, and with a bit set the
GPU must equal the oracle's restatement of the extended state machine, bit for bit.  CPU part: the two
restatements (oracle.cpp, restate_np.py) agree on them; GPU part: parity through the C ABI."""
import math

import numpy as np
import pytest

import sdf_amd as S
from sdf_amd import _ffi
from oracle import oracle as O
from oracle import restate_np as R
from conftest import bits, make_points

ALL = S.EXT_SMOOTH_UNION | S.EXT_BOX_FRAME | S.EXT_CLONE_REPAIR


def smooth_scene(k=1.5, n=9, seed=4):
    rng = np.random.default_rng(seed)
    b = None
    for j in range(n):
        prim = S.SdfSphere(float(rng.uniform(0.6, 1.6))) if j % 2 == 0 else S.SdfBox(tuple(rng.uniform(0.5, 1.5, 3)))
        tr = S.Transform(translation=tuple(rng.uniform(-5, 5, 3)), rotation=S.Quat.from_euler_xyz(*rng.uniform(0, 6.28, 3)))
        leaf = S.SdfBuilder.primitive(prim).transform(tr)
        b = leaf.operation(S.SdfUnion(k)) if b is None else b.with_(leaf)
    return b.finalize().expanded().make_buffer()


def frame_scene():
    a = S.SdfBuilder.primitive(S.SdfBoxFrame((2.0, 1.5, 1.0), 0.2)).transform(
        S.Transform(translation=(1.0, -0.5, 0.5), rotation=S.Quat.from_euler_xyz(0.3, 0.9, 0.2)))
    s = S.SdfBuilder.primitive(S.SdfSphere(0.7)).transform(S.Transform.from_xyz(-4.0, 1.0, 0.0))
    f2 = S.SdfBuilder.primitive(S.SdfBoxFrame((1.0, 1.0, 3.0), 0.35)).transform(S.Transform.from_xyz(3.0, 4.0, -2.0))
    return a.operation(S.SdfUnion(0.0)).with_(s).with_(f2).finalize().expanded().make_buffer()


def clone_scene():
    cl = S.SdfBuilder.primitive(S.SdfSphere(1.0)).operation(S.SdfCaaClone((6, 5, 7), (-1, -2, 0), (2, 1, 1)))
    far = S.SdfBuilder.primitive(S.SdfBox((1, 2, 1))).transform(S.Transform.from_xyz(25.0, 0.0, 0.0))
    return cl.operation(S.SdfUnion(0.0)).with_(far).finalize().expanded().make_buffer()


def mixed_scene():
    cl = S.SdfBuilder.primitive(S.SdfBoxFrame((1.0, 1.0, 1.0), 0.15)).operation(S.SdfCaaClone((5, 5, 5), (-1, -1, -1), (1, 1, 1)))
    a = S.SdfBuilder.primitive(S.SdfSphere(2.0)).transform(S.Transform.from_xyz(0.0, 9.0, 0.0))
    b = S.SdfBuilder.primitive(S.SdfBox((1.0, 3.0, 1.0))).transform(S.Transform.from_xyz(0.5, 6.5, 0.5))
    return cl.operation(S.SdfUnion(1.0)).with_(a).with_(b).finalize().expanded().make_buffer()


SCENES = {"smooth": (smooth_scene, S.EXT_SMOOTH_UNION, 8.0), "frame": (frame_scene, S.EXT_BOX_FRAME, 8.0),
          "clone": (clone_scene, S.EXT_CLONE_REPAIR, 20.0), "mixed": (mixed_scene, ALL, 12.0)}


# ---- CPU: defaults are as written, and the two restatements agree on the extensions ----------------
def test_default_is_as_written():
    rc, info, msg = S.compact_host(frame_scene())
    assert rc == _ffi.ERR_UNSUPPORTED_OP and "primitive op code: 5" in msg          # faux_shader.rs:65 panic
    rc, info, _ = S.compact_host(smooth_scene())
    assert rc == 0 and info.extensions == 0 and info.bounds == S.BOUNDS_BVH          # smooth_radius ignored (:87)
    rc, info, _ = S.compact_host(smooth_scene(), extensions=S.EXT_SMOOTH_UNION)
    assert rc == 0 and info.extensions == S.EXT_SMOOTH_UNION and info.bounds == S.BOUNDS_NONE
    rc, info, _ = S.compact_host(smooth_scene(k=0.0), extensions=ALL)
    assert rc == 0 and info.extensions == 0 and info.bounds == S.BOUNDS_BVH          # nothing to extend: pruned path
    rc, _, msg = S.compact_host(smooth_scene(), bounds=S.BOUNDS_BVH, extensions=S.EXT_SMOOTH_UNION)
    assert rc == _ffi.ERR_BAD_TREE
    rc, _, _ = S.compact_host(smooth_scene(), extensions=64)
    assert rc == _ffi.ERR_INVALID_ARG


def test_which_walk_serves_a_tree():
    # sdfgpu_tree_info.walk: small trees under an extension walk exhaustively, from 256 resident nodes on they prune,
    # unless the smooth radius is of the order of the scene; plain buffers take the hierarchy, the crate's literal
    # buffers (PCA trans_inverse in the merged boxes) the chain path
    rc, info, _ = S.compact_host(smooth_scene(), extensions=S.EXT_SMOOTH_UNION)
    assert rc == 0 and info.walk == S.WALK_GENERAL_EXHAUSTIVE
    rc, info, _ = S.compact_host(smooth_scene(k=0.3, n=200), extensions=S.EXT_SMOOTH_UNION)
    assert rc == 0 and info.walk == S.WALK_GENERAL_PRUNED
    rc, info, _ = S.compact_host(smooth_scene(k=40.0, n=200), extensions=S.EXT_SMOOTH_UNION)
    assert rc == 0 and info.walk == S.WALK_GENERAL_EXHAUSTIVE
    rc, info, _ = S.compact_host(smooth_scene(k=0.3, n=200))
    assert rc == 0 and info.walk == S.WALK_HIERARCHY and info.bounds == S.BOUNDS_BVH
    rc, info, _ = S.compact_host(S.random_scene_buffer(300, 5, opts={"pca_trans_inverse": True}))
    assert rc == 0 and info.walk == S.WALK_CHAIN


@pytest.mark.parametrize("name", list(SCENES))
def test_two_restatements_agree_on_extensions(name):
    make, ext, extent = SCENES[name]
    buf = make()
    pts = make_points(40, extent, 3)
    old = R.set_extensions(ext)
    try:
        with O.extensions(ext):
            want, _ = O.nearest(buf, pts, O.MODE_FAUX)
            wex, _ = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
        got = np.array([R.faux_nearest_scalar(buf, p) for p in pts], np.float32)
        gex = np.array([R.faux_nearest_scalar(buf, p, prune=False) for p in pts], np.float32)
    finally:
        R.set_extensions(old)
    assert np.array_equal(bits(got), bits(want))      # the state machine with its leaf-pair prune
    assert np.array_equal(bits(gex), bits(wex))       # and without it: the contract
    assert np.all(np.isfinite(wex))
    # The as-written leaf-pair prune (faux_shader.rs:117-126) is only sound for plain-min combines over
    # boxes that bound their subtree.  A smooth union lowers values below the minimum, and a clone node
    # carries its CHILD's box (node.rs:43-49, SURVEY C-4), not the lattice's, so the prune cuts the copies in
    # other cells.  The exhaustive walk is the contract under extensions; only the box frame keeps both equal.
    if name == "frame":
        assert np.array_equal(bits(want), bits(wex))


def test_smooth_union_is_a_smooth_minimum():
    buf = smooth_scene(k=2.0)
    pts = make_points(500, 8.0, 5)
    plain, _ = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
    with O.extensions(S.EXT_SMOOTH_UNION):
        sm, _ = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
    assert np.all(sm <= plain) and np.any(sm < plain)
    assert np.all(plain - sm <= 8 * 2.0 / 4 + 1e-5)     # at most k/4 per union level


def test_repaired_clone_repeats_the_primitive():
    buf = clone_scene()
    prim_ops = buf.downtree_buffer["op_code"][buf.downtree_buffer["is_primitive"] == 1]
    sphere = int(np.flatnonzero(prim_ops == _ffi.PRIM_SPHERE)[0])      # pre-order ordinal of the cloned sphere
    with O.extensions(S.EXT_CLONE_REPAIR):
        # lattice cells (ix, iy, iz) inside the limits hold a copy of the unit sphere
        for cell in [(0, 0, 0), (2, 1, 1), (-1, -2, 0), (1, -1, 1)]:
            c = np.array([cell[0] * 6.0, cell[1] * 5.0, cell[2] * 7.0, 1.0], np.float32)
            d, i = O.nearest(buf, c[None, :], O.MODE_EXHAUSTIVE)
            assert d[0] == np.float32(-1.0) and i[0] == sphere
    d, _ = O.nearest(buf, np.array([[0, 0, 0, 1]], np.float32), O.MODE_EXHAUSTIVE)
    assert d[0] > 10.0                                   # as written: the clone subtree is +inf


# ---- GPU parity -------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_gpu_extensions_match_oracle(gpu, name):
    make, ext, extent = SCENES[name]
    buf = make()
    info = gpu.upload_tree(buf, extensions=ext)
    assert info.extensions == ext and info.bounds == S.BOUNDS_NONE
    pts = make_points(3000, extent, 8)
    with O.extensions(ext):
        d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE, threads=4)
        kd0, ki0 = O.knn(buf, pts[:400], 3)
    d, i = gpu.nearest(pts)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    kd, ki = gpu.knn(pts[:400], 3)
    assert np.array_equal(bits(kd), bits(kd0)) and np.array_equal(ki, ki0)
    w, h = 96, 64
    cam = S.camera_look_at((0.3 * extent, 0.2 * extent, -2.2 * extent), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
    mp = S.march_params(max_steps=96, eps_hit=1e-3, t_min=0.0, t_max=8 * extent, normal_h=1e-3)
    with O.extensions(ext):
        ref = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    img = gpu.render(cam, w, h, mp)
    assert ref["hit"].sum() > 20
    assert np.array_equal(img["hit"], ref["hit"]) and np.array_equal(img["prim"], ref["prim"])
    assert np.array_equal(bits(img["depth"]), bits(ref["depth"])) and np.array_equal(bits(img["normal"]), bits(ref["normal"]))


@pytest.mark.gpu
def test_gpu_default_ignores_the_extension_data(gpu):
    # same buffers, no extension bit: smooth_radius is ignored and the clone subtree is +inf, as the crate is written
    for make in (smooth_scene, clone_scene):
        buf = make()
        info = gpu.upload_tree(buf)
        assert info.extensions == 0
        pts = make_points(2000, 12.0, 2)
        d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
        d, i = gpu.nearest(pts)
        assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)

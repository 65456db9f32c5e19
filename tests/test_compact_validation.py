"""sdfgpu_compact_host: the host half of upload (validation + compaction).
Every reference panic!/assert! site on this path must come back as an error
code, never as a fault (SURVEY 8b "Errors")."""
import numpy as np
import pytest

import sdf_amd as S
from sdf_amd import _ffi


def good(n=9, seed=4):
    return S.random_scene_buffer(n, seed)


def mutate(buf, fn):
    dt, ut = buf.downtree_buffer.copy(), buf.uptree_buffer.copy()
    fn(dt, ut)
    return S.SdfTreeBuffer(dt, ut)


def test_accepts_well_formed_trees_and_reports_info():
    for n in (1, 2, 3, 100):
        buf = good(n, n)
        rc, info, msg = S.compact_host(buf)
        assert rc == 0, msg
        assert (info.n_nodes, info.n_prims) == (2 * n - 1, n)
        assert info.bounds == S.BOUNDS_BVH and info.identity_inner == 1 and info.rigid_leaves == 1
        assert info.max_level == int(np.ceil(np.log2(n))) + 1 if n > 1 else info.max_level == 1
        assert info.device_bytes == 64 * (n + info.n_inner)
    rc, info, _ = S.compact_host(S.SdfTreeBuffer.make_empty())
    assert rc == 0 and info.n_nodes == 0


@pytest.mark.parametrize("bounds", [S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE, S.BOUNDS_BVH])
def test_every_bounds_mode_compacts(bounds):
    rc, info, msg = S.compact_host(good(50, 2), bounds=bounds)
    assert rc == 0 and info.bounds == bounds, msg
    assert info.device_bytes == 64 * 50 + (128 if bounds == S.BOUNDS_OBB else 64) * 49


def test_scene_aabb_contains_all_primitive_centres():
    buf = good(200, 6)
    rc, info, _ = S.compact_host(buf)
    prims = buf.downtree_buffer[buf.downtree_buffer["is_primitive"] == 1]
    c = prims["bounding_box"]["matrix"][:, 12:15]
    assert np.all(c >= np.array(info.scene_lo)) and np.all(c <= np.array(info.scene_hi))


CASES = {
    "bad_layout": (lambda d, u: None, _ffi.ERR_LAYOUT),
    "root_level": (lambda d, u: d["level"].__setitem__(0, 2), _ffi.ERR_BAD_TREE),
    "root_len": (lambda d, u: d["len"].__setitem__(0, 3), _ffi.ERR_BAD_TREE),
    "len_overrun": (lambda d, u: d["len"].__setitem__(len(d) - 1, 5), _ffi.ERR_BAD_TREE),
    "prim_with_len": (lambda d, u: d["len"].__setitem__(int(np.flatnonzero(d["is_primitive"])[0]), 1), _ffi.ERR_BAD_TREE),
    "child_level": (lambda d, u: d["level"].__setitem__(1, 5), _ffi.ERR_BAD_TREE),
    "level_zero": (lambda d, u: d["level"].__setitem__(3, 0), _ffi.ERR_BAD_TREE),
    "too_deep": (lambda d, u: d["level"].__setitem__(3, 300), _ffi.ERR_TOO_DEEP),
    "bool_garbage": (lambda d, u: d["is_primitive"].__setitem__(2, 7), _ffi.ERR_BAD_TREE),
    "unknown_primitive": (lambda d, u: d["op_code"].__setitem__(int(np.flatnonzero(d["is_primitive"])[0]), 5), _ffi.ERR_UNSUPPORTED_OP),
    "clone_op": (lambda d, u: d["op_code"].__setitem__(0, 1), _ffi.ERR_UNSUPPORTED_OP),
    "nan_transform": (lambda d, u: d["bounding_box"]["trans_inverse"][2].__setitem__(5, np.nan), _ffi.ERR_NON_FINITE),
    "inf_radius": (lambda d, u: [d["op_specific"]["floats"][i].__setitem__(0, np.inf) for i in np.flatnonzero((d["is_primitive"] == 1) & (d["op_code"] == 0))[:1]], _ffi.ERR_NON_FINITE),
    "uptree_mismatch": (lambda d, u: u["level"].__setitem__(0, 9), _ffi.ERR_BAD_TREE),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_malformed_buffers_are_rejected_with_a_code(name):
    fn, want = CASES[name]
    buf = mutate(good(), fn)
    rc, info, msg = S.compact_host(buf, layout=7 if name == "bad_layout" else _ffi.LAYOUT_GLAM_SSE2)
    assert rc == want, (name, rc, msg)
    assert msg


def test_three_children_rejected():
    # frames hold two results (faux_shader.rs:14-15): a 3-child union would index out of bounds
    buf = good(3, 1)   # root(U) -> [U -> [p, p], p] or similar; rebuild as root with 3 leaf children
    dt, ut = buf.downtree_buffer.copy(), buf.uptree_buffer.copy()
    prims = dt[dt["is_primitive"] == 1].copy()
    root = dt[:1].copy()
    root["len"] = 3
    prims["level"] = 2
    ndt = np.concatenate([root, prims])
    nut = np.zeros(4, _ffi.UT_DTYPE)
    nut["level"] = [2, 2, 2, 1]
    nut["op_code"] = list(prims["op_code"]) + [0]
    rc, _, msg = S.compact_host(S.SdfTreeBuffer(ndt, nut))
    assert rc == _ffi.ERR_BAD_TREE and "two children" in msg


def test_non_identity_inner_transform_takes_the_chain_or_the_general_path():
    # SURVEY C-3: the crate's literal merged boxes re-base the point.  Rigid, affine re-basing (what obb.rs:183-185
    # writes) is pruned over the composed maps' world boxes with the per-level chain kept for the leaf values
    # (chain path, round 3); asked for without pruning, the buffer is walked exhaustively (general path).
    buf = S.random_scene(9, 3, opts={"pca_trans_inverse": True}).expanded().make_buffer()
    rc, info, msg = S.compact_host(buf)
    assert rc == 0 and info.bounds == S.BOUNDS_BVH and info.identity_inner == 0, msg
    n_prims = int(buf.downtree_buffer["is_primitive"].sum())
    assert info.device_bytes > 64 * (2 * n_prims - 1) + 8 * n_prims    # leaves + hierarchy + per-primitive chain index + chain records
    rc, info, msg = S.compact_host(buf, bounds=S.BOUNDS_NONE)
    assert rc == 0 and info.bounds == S.BOUNDS_NONE and info.identity_inner == 0, msg
    assert info.device_bytes == 192 * buf.buffer_len    # one three-line GpuGenNode per node
    # a re-basing node that is not an isometry cannot be pruned over world boxes: exhaustive
    dt = buf.downtree_buffer.copy()
    inner = int(np.flatnonzero(dt["is_primitive"] == 0)[1])
    dt["bounding_box"]["trans_inverse"][inner][0] *= np.float32(1.5)
    rc, info, msg = S.compact_host(S.SdfTreeBuffer(dt, buf.uptree_buffer))
    assert rc == 0 and info.bounds == S.BOUNDS_NONE, msg
    for b in (S.BOUNDS_AABB, S.BOUNDS_OBB):
        rc, _, msg = S.compact_host(buf, bounds=b)
        assert rc == _ffi.ERR_BAD_TREE and "identity" in msg


def test_general_path_depth_limit():
    buf = S.random_scene(9, 3, opts={"pca_trans_inverse": True}).expanded().make_buffer()
    dt, ut = buf.downtree_buffer.copy(), buf.uptree_buffer.copy()
    # a 40-level chain of one-child unions above the tree
    chain = np.zeros(40, _ffi.DT_DTYPE)
    chain["level"] = np.arange(1, 41)
    chain["len"] = len(dt) + 39 - np.arange(40)
    chain["bounding_box"] = dt["bounding_box"][0]
    chain["parent_is_union"] = 1
    chain["parent_is_union"][0] = 0
    dt["level"] += 40
    dt["parent_is_union"][0] = 1
    ut["level"] += 40
    uchain = np.zeros(40, _ffi.UT_DTYPE)
    uchain["level"] = np.arange(40, 0, -1)
    big = S.SdfTreeBuffer(np.concatenate([chain, dt]), np.concatenate([ut, uchain]))
    rc, _, msg = S.compact_host(big, bounds=S.BOUNDS_NONE)
    assert rc == _ffi.ERR_TOO_DEEP and "general path" in msg
    rc, info, msg = S.compact_host(big)       # the chain path holds 64 re-basing levels and builds its own hierarchy
    assert rc == 0 and info.bounds == S.BOUNDS_BVH, msg


def test_non_rigid_leaf_falls_back_to_no_pruning():
    def scale_leaf(d, u):
        i = int(np.flatnonzero(d["is_primitive"])[0])
        d["bounding_box"]["trans_inverse"][i][:3] *= 2.0
    buf = mutate(good(), scale_leaf)
    rc, info, msg = S.compact_host(buf)
    assert rc == 0 and info.bounds == S.BOUNDS_NONE and info.rigid_leaves == 0
    for mode in (S.BOUNDS_AABB, S.BOUNDS_BVH, S.BOUNDS_OBB):   # every pruned mode compares world boxes with leaf values
        rc, _, msg = S.compact_host(buf, bounds=mode)
        assert rc == _ffi.ERR_BAD_TREE and "rigid" in msg


def test_obb_mode_requires_affine_boxes():
    buf = mutate(good(), lambda d, u: d["bounding_box"]["scale"][1].__setitem__(3, 1.0))
    assert S.compact_host(buf, bounds=S.BOUNDS_AABB)[0] == 0
    assert S.compact_host(buf, bounds=S.BOUNDS_OBB)[0] == _ffi.ERR_BAD_TREE


def deep_chain(depth):
    """union(p0, union(p1, union(p2, ...))): a buffer `depth` levels deep, as the crate's builder nests it."""
    node = S.SdfBuilder.primitive(S.SdfSphere(0.5)).transform(S.Transform.from_xyz(3.0 * depth, 0, 0))
    for k in range(depth - 1, -1, -1):
        leaf = S.SdfBuilder.primitive(S.SdfBox((0.4, 0.6, 0.5))).transform(S.Transform.from_xyz(3.0 * k, 0.5 * (k % 3), 0))
        node = leaf.operation(S.SdfUnion(0.0)).with_(node)
    return node.finalize().expanded().make_buffer()


def test_deep_buffer_is_accepted_when_the_library_builds_the_hierarchy():
    # the reference's stack holds 256 frames (faux_shader.rs:100); only modes that walk the buffer's own
    # topology are limited by the 64-entry wave stack
    buf = deep_chain(100)
    assert int(buf.downtree_buffer["level"].max()) > 64
    rc, info, msg = S.compact_host(buf)
    assert rc == 0 and info.bounds == S.BOUNDS_BVH and info.stack_depth <= 64, msg
    for mode in (S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE):
        rc, _, msg = S.compact_host(buf, bounds=mode)
        assert rc == _ffi.ERR_TOO_DEEP, msg

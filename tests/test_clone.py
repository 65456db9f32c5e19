"""SdfCaaClone (operation op 1; elements.rs:215-262, faux_shader.rs:75-78,90).

As the flat query is written, uptree op 1 returns the RIGHT branch distance while a one-slot
node only ever fills the left one, so a clone subtree evaluates to +inf (SURVEY A-6).  That is
restated, not repaired: the oracle runs the literal state machine, the library validates and
numbers the subtree and leaves it out of the resident tree.  The crate's recursive n-ary query
(node.rs:331-368) does descend through a clone (downtree_transform), and so does the host mirror.
"""
import numpy as np
import pytest

import sdf_amd as S
from sdf_amd import _ffi
from oracle import oracle as O
from conftest import bits, make_points


def clone_scene():
    """union( clone(sphere r=1), sphere r=1 at (100,0,0), box at (0,30,0) )"""
    cl = S.SdfBuilder.primitive(S.SdfSphere(1.0)).operation(S.SdfCaaClone((10, 10, 10), (-1, -1, -1), (1, 1, 1)))
    far = S.SdfBuilder.primitive(S.SdfSphere(1.0)).transform(S.Transform.from_translation((100.0, 0.0, 0.0)))
    box = S.SdfBuilder.primitive(S.SdfBox((1, 2, 3))).transform(S.Transform.from_translation((0.0, 30.0, 0.0)))
    return cl.operation(S.SdfUnion(0.0)).with_(far).with_(box).finalize()


def test_clone_block_as_make_buffer_writes_it():
    node = clone_scene()
    buf = node.expanded().make_buffer()
    dt = buf.downtree_buffer
    cl = np.flatnonzero((dt["is_primitive"] == 0) & (dt["op_code"] == _ffi.OP_CAA_CLONE))
    assert len(cl) == 1
    b = dt[cl[0]]
    assert b["len"] == 1 and dt[cl[0] + 1]["is_primitive"] == 1 and dt[cl[0] + 1]["level"] == b["level"] + 1
    v = b["op_specific"]["vec4s"]
    assert np.array_equal(v[0], [10, 10, 10, 0]) and np.array_equal(v[1], [-1, -1, -1, 0]) and np.array_equal(v[2], [1, 1, 1, 0])
    assert buf.n_prims == 3


def test_flat_query_makes_a_clone_subtree_infinite_as_written():
    buf = clone_scene().expanded().make_buffer()
    dt = buf.downtree_buffer
    pts = make_points(400, 40.0, 9)
    pts[0, :3] = (0.0, 0.0, 0.0)            # centre of the cloned sphere: it would be -1 if it counted
    # primitive ordinals of the two live primitives
    live = [k for k, i in enumerate(np.flatnonzero(dt["is_primitive"] == 1))
            if not (dt[i - 1]["op_code"] == _ffi.OP_CAA_CLONE and dt[i - 1]["is_primitive"] == 0)]
    assert len(live) == 2
    want = np.min(np.stack([O.eval_prim(buf, k, pts) for k in live]), axis=0)
    for mode in (O.MODE_FAUX, O.MODE_EXHAUSTIVE, O.MODE_TUNED, O.MODE_FAUX_FAITHFUL):
        d, i = O.nearest(buf, pts, mode=mode)
        assert np.array_equal(bits(d), bits(want)), mode
        assert set(np.unique(i)) <= set(live)
    d8, i8 = O.knn(buf, pts, 4)
    assert np.all(np.isinf(d8[:, 2:])) and np.all(i8[:, 2:] == S.NO_PRIM)
    assert np.array_equal(bits(d8[:, 0]), bits(want))


def test_clone_only_tree_is_infinite():
    node = S.SdfBuilder.primitive(S.SdfSphere(2.0)).operation(S.SdfCaaClone((5, 5, 5), (0, 0, 0), (3, 3, 3))).finalize()
    buf = node.expanded().make_buffer()
    d, i = O.nearest(buf, make_points(50, 10.0, 1), mode=O.MODE_FAUX)
    assert np.all(np.isinf(d)) and np.all(i == S.NO_PRIM)
    rc, info, msg = S.compact_host(buf)
    assert rc == 0 and info.n_prims == 1 and info.n_inner == 0, msg


def test_compaction_leaves_clone_subtrees_out():
    buf = clone_scene().expanded().make_buffer()
    for bounds in (S.BOUNDS_AUTO, S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE, S.BOUNDS_BVH):
        rc, info, msg = S.compact_host(buf, bounds=bounds)
        assert rc == 0, msg
        assert info.n_prims == 3 and info.n_inner == 1        # one resident inner node over the two live leaves
        assert info.identity_inner == 1                        # the clone's own transform does not force the general path


def test_clone_with_two_slots_is_rejected():
    buf = S.random_scene(4, 2).expanded().make_buffer()
    dt, ut = buf.downtree_buffer.copy(), buf.uptree_buffer.copy()
    dt["op_code"][0] = _ffi.OP_CAA_CLONE
    ut["op_code"][-1] = _ffi.OP_CAA_CLONE
    rc, _, msg = S.compact_host(S.SdfTreeBuffer(dt, ut))
    assert rc == _ffi.ERR_UNSUPPORTED_OP and "slots" in msg


def test_recursive_query_descends_through_a_clone():
    # node.rs:358: the child's downtree transform (elements.rs:234-241, fmod of p - d/2) is applied on the way down
    node = clone_scene()
    d, i = node.nearest_neighbor((1.0, 2.0, 3.0))
    # clone child sorted first (nearest box); lattice point fmod((1,2,3) - 5, 10) = (-4,-3,-2); sphere r = 1
    assert i == 0
    assert abs(d - (np.sqrt(29.0) - 1.0)) < 1e-5

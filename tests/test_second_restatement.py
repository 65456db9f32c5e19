"""A second CPU restatement of the flat query (oracle/restate_np.py: numpy float32, one rounded
operation per line, written from sdf/src/faux_shader.rs directly) against the first
(oracle/oracle.cpp), bit for bit, on the committed golden scenes and on the general / clone cases.

This does not pin parity to the reference (it holds no fixtures and cannot be built here: parity
stays "unpinned"); it removes the risk of a single restatement misreading the
source unnoticed.
"""
import os

import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from oracle import restate_np as R
from conftest import bits, make_points
from golden_util import GOLDEN, load

IDS = [os.path.basename(p)[:-4] for p in GOLDEN]


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_leaf_chain_matches_golden_and_oracle(path):
    z, buf, _, _ = load(path)
    d, i = R.nearest_exhaustive(buf, z["pts"])
    assert np.array_equal(bits(d), bits(z["dist"])), "distance bits differ from the golden vectors"
    assert np.array_equal(i, z["prim"]), "indices differ from the golden vectors"
    d0, i0 = O.nearest(buf, z["pts"], O.MODE_EXHAUSTIVE)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_state_machine_as_written_matches_oracle(path):
    # faux_shader.rs:96-208 literally (frames, prune test with MINDIST / MAXDIST, up-pass), point by point
    z, buf, _, _ = load(path)
    pts = z["pts"][:24]
    want, _ = O.nearest(buf, pts, O.MODE_FAUX)
    got = np.array([R.faux_nearest_scalar(buf, p) for p in pts], np.float32)
    assert np.array_equal(bits(got), bits(want))


@pytest.mark.parametrize("path", [p for p in GOLDEN if "random_7" in p or "duplicates" in p])
def test_bounds_match_oracle(path):
    # MINDIST (faux_shader.rs:54-58) and MAXDIST (:39-52) of every node's box
    z, buf, _, _ = load(path)
    pts = z["pts"][:64]
    for node in range(buf.buffer_len):
        lo0, hi0 = O.bounds(buf, node, pts)
        box = buf.downtree_buffer[node]["bounding_box"]
        assert np.array_equal(bits(R.mindist(box, pts)), bits(lo0)), f"mindist, node {node}"
        assert np.array_equal(bits(R.maxdist(box, pts)), bits(hi0)), f"maxdist, node {node}"


@pytest.mark.parametrize("n", [3, 9, 40])
def test_general_chain_with_internal_transforms(n):
    # buffers as the crate itself would write them: PCA trans_inverse on merged boxes (SURVEY C-3),
    # so the per-level point chain is not the identity
    buf = S.random_scene(n, 0x6E00 + n, opts={"pca_trans_inverse": True}).expanded().make_buffer()
    pts = make_points(200, S.default_extent(n), n)
    d, i = R.nearest_exhaustive(buf, pts)
    d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    got = np.array([R.faux_nearest_scalar(buf, p) for p in pts[:12]], np.float32)
    want, _ = O.nearest(buf, pts[:12], O.MODE_FAUX)
    assert np.array_equal(bits(got), bits(want))


def test_clone_subtree_as_written():
    # SdfCaaClone: downtree op 1 computes the lattice point, uptree op 1 returns the right branch,
    # which a one-slot node never fills (faux_shader.rs:75-78,90): +inf for the subtree
    cl = S.SdfBuilder.primitive(S.SdfSphere(1.0)).operation(S.SdfCaaClone((10, 10, 10), (-1, -1, -1), (1, 1, 1)))
    node = (S.SdfBuilder.primitive(S.SdfSphere(2.0)).transform(S.Transform.from_xyz(30, 0, 0))
            .operation(S.SdfUnion(0.0)).with_(cl).finalize())
    buf = node.expanded().make_buffer()
    pts = make_points(64, 40.0, 11)
    want, _ = O.nearest(buf, pts, O.MODE_FAUX)
    got = np.array([R.faux_nearest_scalar(buf, p) for p in pts], np.float32)
    assert np.array_equal(bits(got), bits(want))
    d, _ = R.nearest_exhaustive(buf, pts)
    assert np.array_equal(bits(d), bits(want))


def test_reference_result_beyond_the_query_domain_is_recorded():
    """Recorded deviation (sdfgpu.h "Domain of a query point"): for a FINITE point with a
    coordinate beyond 2^62 the library returns (NaN, SDF_NO_PRIM); the reference's literal fold returns whatever its
    overflowed arithmetic leaves.  This test keeps that literal result visible instead of restating the library's
    rule on both sides: both restatements run the state machine of faux_shader.rs:96-208 as written at such points
    and must agree with each other; the value is +inf or NaN (inf - inf after an overflowing product), or, where
    nothing overflowed yet, an ordinary huge distance -- never something the library's rule would contradict inside
    the domain."""
    import warnings
    buf = S.random_scene_buffer(9, 0xD0)
    pts = np.array([[3e38, 0, 0, 1], [0, -2e30, 5e18, 1], [1e25, 1e25, -1e25, 1], [5e18, 5e18, 5e18, 1]], np.float32)
    assert (np.abs(pts[:, :3]).max(axis=1) > 2.0 ** 62).all()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with np.errstate(all="ignore"):
            want = [R.faux_nearest_scalar(buf, p) for p in pts]
    got, _ = O.nearest(buf, pts, O.MODE_FAUX)
    for d, g in zip(want, got):
        d = np.float32(d)
        assert (np.isnan(d) and np.isnan(g)) or bits(np.array([d]))[0] == bits(np.array([g]))[0]
        assert np.isnan(g) or np.isinf(g) or g > 2.0 ** 61

"""The pruned walks of the general path (kernels.hip walk_eval_pruned / walk_general_pruned: trees in which an opt-in
extension is in effect -- smooth unions, repaired clones, box frames -- with rigid maps throughout) against the
exhaustive walks of the same library (SDFGPU_GEN_PRUNE=0 at upload) on nearest, kNN and rendered frames, bit for bit,
and against the oracle's exhaustive restatement on a sample.  Scenes of 150 ... 4000 primitives: pruning is only
switched on from 256 resident nodes, and not where the smooth radii are of the order of the scene (k = 4 case: both
uploads then run the exhaustive walk, which the test passes trivially -- kept as the check of that decision)."""
import math
import os
import time

import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits, make_points


def big_smooth_scene(n, k, seed, clone=False, frame=False):
    rng = np.random.default_rng(seed)
    ext = 5.0 * (n / 9.0) ** (1.0 / 3.0)
    b = None
    for j in range(n):
        r = j % 3
        if frame and r == 2:
            prim = S.SdfBoxFrame(tuple(rng.uniform(0.8, 1.6, 3)), 0.2)
        elif r == 0:
            prim = S.SdfSphere(float(rng.uniform(0.6, 1.6)))
        else:
            prim = S.SdfBox(tuple(rng.uniform(0.5, 1.5, 3)))
        tr = S.Transform(translation=tuple(rng.uniform(-ext, ext, 3)), rotation=S.Quat.from_euler_xyz(*rng.uniform(0, 6.28, 3)))
        leaf = S.SdfBuilder.primitive(prim).transform(tr)
        if clone and j == 1:
            leaf = S.SdfBuilder.primitive(S.SdfSphere(0.7)).operation(S.SdfCaaClone((4, 5, 3), (-1, -2, 0), (2, 1, 1)))
        b = leaf.operation(S.SdfUnion(k)) if b is None else b.with_(leaf)
    return b.finalize().expanded().make_buffer(), ext


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,ext_bits,clone,frame", [
    (150, 0.5, S.EXT_SMOOTH_UNION, False, False),
    (700, 0.8, S.EXT_SMOOTH_UNION, False, False),
    (700, 4.0, S.EXT_SMOOTH_UNION, False, False),
    (300, 0.0, S.EXT_CLONE_REPAIR, True, False),
    (300, 1.0, S.EXT_SMOOTH_UNION | S.EXT_CLONE_REPAIR | S.EXT_BOX_FRAME, True, True),
    (500, 0.0, S.EXT_BOX_FRAME, False, True),
    (4000, 0.8, S.EXT_SMOOTH_UNION, False, False),
    (4000, 0.0, S.EXT_BOX_FRAME, False, True),
])
def test_pruned_general_walks(gpu, n, k, ext_bits, clone, frame):
    buf, ext = big_smooth_scene(n, k, 11 + n, clone, frame)
    pts = make_points(6000, ext, 5, scale=1.4)
    w, h = 128, 96
    cam = S.camera_look_at((0.3 * ext, 0.2 * ext, -2.4 * ext), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
    mp = S.march_params(max_steps=96, eps_hit=1e-3, t_min=0.0, t_max=8 * ext, normal_h=1e-3)
    res = {}
    for mode in ("0", "1"):
        os.environ["SDFGPU_GEN_PRUNE"] = mode
        info = gpu.upload_tree(buf, extensions=ext_bits)
        assert info.extensions == ext_bits and info.bounds == S.BOUNDS_NONE
        t0 = time.perf_counter()
        d, i = gpu.nearest(pts)
        t1 = time.perf_counter()
        kd, ki = gpu.knn(pts[:1500], 5)
        img = gpu.render(cam, w, h, mp)
        t2 = time.perf_counter()
        res[mode] = (d, i, kd, ki, img, t1 - t0, t2 - t1)
    os.environ.pop("SDFGPU_GEN_PRUNE")
    a, b = res["0"], res["1"]
    print(f"n={n} k={k}: nearest {a[5]*1e3:.1f} -> {b[5]*1e3:.1f} ms, knn+render {a[6]*1e3:.1f} -> {b[6]*1e3:.1f} ms, hits {int(a[4]['hit'].sum())}")
    assert np.array_equal(bits(a[0]), bits(b[0])), np.flatnonzero(bits(a[0]) != bits(b[0]))[:10]
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(bits(a[2]), bits(b[2])) and np.array_equal(a[3], b[3])
    for key in ("hit", "prim"):
        assert np.array_equal(a[4][key], b[4][key]), key
    for key in ("depth", "normal"):
        assert np.array_equal(bits(a[4][key]), bits(b[4][key])), key
    assert a[4]["hit"].sum() > 20
    with O.extensions(ext_bits):
        d0, i0 = O.nearest(buf, pts[:800], O.MODE_EXHAUSTIVE, threads=8)
    assert np.array_equal(bits(b[0][:800]), bits(d0)) and np.array_equal(b[1][:800], i0)

#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small tree buffers (crate record layout), query
points, a tiny camera, and the outputs of the CPU oracle on them.

The reference (Rust) cannot run here and holds no fixtures of its own
(SURVEY 8c), so these vectors come from oracle/oracle.cpp, the line-by-line
restatement of sdf/src/faux_shader.rs, cross-checked in tests/test_oracle.py
against hand-derived answers and a float64 numpy evaluation.  They pin the
oracle against regressions (CPU test) and the kernels against the oracle
without needing the oracle at run time (GPU test).

Run from the repo root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sdf_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def points(n, extent, seed):
    rng = np.random.default_rng(seed)
    p = np.ones((n, 4), np.float32)
    p[:, :3] = rng.uniform(-1.2 * extent, 1.2 * extent, (n, 3)).astype(np.float32)
    return p


def duplicate_scene():
    t = S.Transform.from_translation((1.0, 2.0, 3.0))
    r = S.Transform(translation=(-4.0, 1.0, 0.5), rotation=S.Quat.from_euler_xyz(0.4, 1.0, 2.0))
    b = (S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t).operation(S.SdfUnion(0.0))
         .with_(S.SdfBuilder.primitive(S.SdfBox((1.0, 0.5, 2.0))).transform(r))
         .with_(S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t))
         .with_(S.SdfBuilder.primitive(S.SdfBox((1.0, 0.5, 2.0))).transform(r)))
    return b.finalize().expanded().make_buffer(), 8.0


def single_sphere():
    b = S.SdfBuilder.primitive(S.SdfSphere(1.0)).transform(S.Transform.from_translation((3.0, -4.0, 5.5)))
    return b.finalize().expanded().make_buffer(), 10.0


def pca_inner_scene():
    """What the crate's own SdfBoundingBox::merge writes (obb.rs:183-185): every merged box carries a PCA
    trans_inverse that re-bases the point handed to its children (SURVEY C-3, 7.4 'one with non-identity
    internal trans_inverse')."""
    n = 12
    return S.random_scene(n, 0x601D0C3, opts={"pca_trans_inverse": True}).expanded().make_buffer(), S.default_extent(n)


def clone_scene():
    """Live primitives mixed with one-slot CAA clones (elements.rs:215-262): as the flat query is written a clone
    subtree contributes +inf (faux_shader.rs:90; SURVEY A-6, 7.4 'one with a CAA-clone')."""
    rng = np.random.default_rng(0xC10E)
    L = 12.0

    def prim():
        e = S.SdfSphere(float(rng.uniform(0.5, 2.0))) if rng.random() < 0.5 else S.SdfBox(tuple(float(v) for v in rng.uniform(0.5, 2.0, 3)))
        return S.SdfBuilder.primitive(e).transform(S.Transform.from_translation(tuple(float(v) for v in rng.uniform(-L, L, 3))))

    parts = [prim() for _ in range(9)]
    for at in (2, 5, 9):
        parts.insert(at, prim().operation(S.SdfCaaClone((6, 6, 6), (-2, -2, -2), (2, 2, 2))))
    b = parts[0].operation(S.SdfUnion(0.0))
    for x in parts[1:]:
        b = b.with_(x)
    return b.finalize().expanded().make_buffer(), L


# bounds modes a fixture can be uploaded with (a buffer with re-basing internal nodes takes AUTO / NONE only)
ALL_BOUNDS = [1, 2, 3, 4]
BOUNDS_OK = {"pca_inner_12": [0, 3, 4]}

SCENES = {
    "pca_inner_12": pca_inner_scene,
    "clone_mix_12": clone_scene,
    "single_sphere": single_sphere,
    "duplicates_tie": duplicate_scene,
    "random_2": lambda: (S.random_scene_buffer(2, 0x601D0002), S.default_extent(2)),
    "random_3": lambda: (S.random_scene_buffer(3, 0x601D0003), S.default_extent(3)),
    "random_7": lambda: (S.random_scene_buffer(7, 0x601D0007), S.default_extent(7)),
    "random_64": lambda: (S.random_scene_buffer(64, 0x601D0040), S.default_extent(64)),
}


def main():
    for name, make in SCENES.items():
        buf, ext = make()
        pts = points(256, ext, len(name))
        # The contract is the exhaustive walk: (min distance, LAST primitive in
        # pre-order attaining it).  The as-written walk returns the same distance;
        # its un-margined leaf-pair prune (faux_shader.rs:119-120, SURVEY C-6) may
        # skip an exactly tying later duplicate, which only shows in the index
        # the build adds (the reference itself returns no index).
        d, i = O.nearest(buf, pts, O.MODE_EXHAUSTIVE)
        d2, i2 = O.nearest(buf, pts, O.MODE_FAUX)
        assert np.array_equal(d.view(np.uint32), d2.view(np.uint32))
        assert np.array_equal(i, i2) or name == "duplicates_tie"
        if name == "clone_mix_12":
            assert len(set(i.tolist())) > 3 and buf.downtree_buffer["op_code"][buf.downtree_buffer["is_primitive"] == 0].max() == 1
        kd, ki = O.knn(buf, pts, 8)
        w, h = 24, 16
        cam = S.camera_look_at((0, 0, -2.5 * ext), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
        mp = S.march_params(64, 1e-3, 0.0, 6 * ext, 1e-4)
        img = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            dt=buf.downtree_buffer.view(np.uint8), ut=buf.uptree_buffer.view(np.uint8),
            pts=pts, dist=d, prim=i, knn_dist=kd, knn_prim=ki,
            cam=np.array(list(cam.eye) + list(cam.origin) + list(cam.du) + list(cam.dv), np.float32),
            march=np.array([mp.max_steps, mp.eps_hit, mp.t_min, mp.t_max, mp.normal_h], np.float64),
            wh=np.array([w, h]), hit=img["hit"], depth=img["depth"], normal=img["normal"], rprim=img["prim"],
            bounds_ok=np.array(BOUNDS_OK.get(name, ALL_BOUNDS)),
        )
        print(name, buf.buffer_len, "nodes;", int(img["hit"].sum()), "hit pixels")


if __name__ == "__main__":
    main()

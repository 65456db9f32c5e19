#!/usr/bin/env python3
"""Primitive parameters a producer should never write (radius / half extent 0, negative, huge, NaN, inf), patched
into an otherwise valid buffer: what upload says, and whether the GPU still agrees with the oracle bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import copy  # noqa: E402

import numpy as np  # noqa: E402

import sdf_amd as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


base = S.random_scene_buffer(60, 0xBADF)
prims = np.flatnonzero(base.downtree_buffer["is_primitive"])
rng = np.random.default_rng(0)
pts = np.ones((4000, 4), np.float32)
pts[:, :3] = rng.uniform(-1.2, 1.2, (4000, 3)) * S.default_extent(60)
gpu = S.SdfGpu(0)
for name, val in [("zero", 0.0), ("negative", -1.5), ("tiny", 1e-30), ("huge", 1e30), ("inf", float("inf")), ("nan", float("nan"))]:
    for kind in (0, 2):   # sphere, box
        buf = copy.deepcopy(base)
        which = [int(i) for i in prims if buf.downtree_buffer[i]["op_code"] == kind][:3]
        for i in which:
            if kind == 0:
                buf.downtree_buffer[i]["op_specific"]["floats"][0] = val
                buf.downtree_buffer[i]["op_specific"]["vec4s"][0][0] = val   # wherever the radius lives
            else:
                buf.downtree_buffer[i]["op_specific"]["vec4s"][0][1] = val
        line = f"{name:9s} {'sphere' if kind == 0 else 'box   '}:"
        for bounds, bn in ((S.BOUNDS_BVH, "bvh"), (S.BOUNDS_AABB, "aabb"), (S.BOUNDS_OBB, "obb"), (S.BOUNDS_NONE, "none")):
            try:
                gpu.upload_tree(buf, bounds=bounds)
            except Exception as e:   # noqa: BLE001
                line += f" {bn}: upload -> {str(e)[:60]!r};"
                continue
            d, i = gpu.nearest(pts)
            d0, i0 = O.nearest(buf, pts, O.MODE_EXHAUSTIVE, threads=8)
            same = np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
            nbad = int(((bits(d) != bits(d0)) | (i != i0)).sum())
            line += f" {bn}: {'same' if same else 'DIFF ' + str(nbad)};"
        print(line)

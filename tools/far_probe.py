#!/usr/bin/env python3
"""Visit counts and frame time for points / a camera far outside the scene (the unclamped copy of the walk;
kernels.hip, aabb_dist2 FAR).  Prints internal nodes visited per wavefront at several distances and the
1080p frame time with the eye 2.5, 20, 60 and 300 scene extents away (narrowing the field of view to match)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdf_amd as S  # noqa: E402
from sdf_amd import _ffi  # noqa: E402

n = 10000
L = S.default_extent(n)
gpu = S.SdfGpu(0)
gpu.upload_tree(S.random_scene_buffer(n, 0x5DF00003))
rng = np.random.default_rng(1)
gpu.set_option(_ffi.OPT_STATS_ASM_WALK, 1)
for f in (0.5, 5.0, 14.0, 20.0, 40.0, 300.0, 1e4):
    d = rng.normal(size=(64000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.ones((64000, 4), np.float32)
    pts[:, :3] = (d * L * f).astype(np.float32)
    t = torch.from_numpy(pts).cuda()
    st = gpu.nearest_stats_dev(t.data_ptr(), 64000)
    print(f"points at {f:g} extents: {st['wave_inner'] / 1000:.1f} internal nodes, {st['wave_leaf'] / 1000:.1f} leaves per wavefront (tree: {n - 1} / {n})")
gpu.set_option(_ffi.OPT_STATS_ASM_WALK, 0)
w, h = 1920, 1080
for f, fov in ((2.5, 0.785), (20.0, 0.1), (60.0, 0.033), (300.0, 0.0066)):
    cam = S.camera_look_at((0.0, 0.0, -f * L), (0.0, 0.0, 0.0), (0, 1, 0), fov, w, h)
    mp = S.march_params(max_steps=128, eps_hit=1e-3, t_min=0.0, t_max=(f + 4.0) * L, normal_h=1e-4)
    outs = [torch.empty(w * h, dtype=torch.uint8, device="cuda"), torch.empty(w * h, dtype=torch.float32, device="cuda"),
            torch.empty(w * h * 3, dtype=torch.float32, device="cuda"), torch.empty(w * h, dtype=torch.int32, device="cuda")]
    for _ in range(3):
        gpu.render_dev(cam, w, h, mp, *(o.data_ptr() for o in outs), 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gpu.render_dev(cam, w, h, mp, *(o.data_ptr() for o in outs), 0)
    torch.cuda.synchronize()
    print(f"eye at {f:g} extents, fov {fov}: {(time.perf_counter() - t0) * 100:.2f} ms per frame, hit fraction {float(outs[0].float().mean()):.3f}")

#!/usr/bin/env python3
"""Device-memory leak probe: contexts created and destroyed, trees re-uploaded, frames and batches on many
streams -- free device memory before and after (hipMemGetInfo through torch)."""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdf_amd as S  # noqa: E402

torch.cuda.init()
buf_a, buf_b = S.random_scene_buffer(2000, 1), S.random_scene_buffer(300, 2)
cam = S.camera_look_at((0, 0, -150.0), (0, 0, 0), (0, 1, 0), 0.8, 320, 200)
mp = S.march_params(max_steps=64, eps_hit=1e-3, t_min=0.0, t_max=600.0, normal_h=1e-4)
pts = np.ones((20000, 4), np.float32)
pts[:, :3] = np.random.default_rng(0).uniform(-60, 60, (20000, 3))


def round_trip():
    g = S.SdfGpu(0)
    for b, bounds in ((buf_a, S.BOUNDS_AUTO), (buf_b, S.BOUNDS_OBB), (buf_a, S.BOUNDS_AABB), (buf_b, S.BOUNDS_NONE)):
        g.upload_tree(b, bounds=bounds)
        g.nearest(pts)
        g.knn(pts[:3000], 8)
        g.render(cam, 320, 200, mp)
    streams = [torch.cuda.Stream() for _ in range(20)]   # more than the 16 scratch slots
    outs = [torch.empty(320 * 200, dtype=torch.uint8, device="cuda"), torch.empty(320 * 200, dtype=torch.float32, device="cuda"),
            torch.empty(320 * 200 * 3, dtype=torch.float32, device="cuda"), torch.empty(320 * 200, dtype=torch.int32, device="cuda")]
    for s in streams:
        g.render_dev(cam, 320, 200, mp, *(o.data_ptr() for o in outs), s.cuda_stream)
    torch.cuda.synchronize()
    g.close() if hasattr(g, "close") else None
    del g, streams, outs


def free_now():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return torch.cuda.mem_get_info()[0]


def measure(blocks=4, per_block=25):
    """Free device memory after 1 round trip and after each further block; returns (marks, MiB lost per block
    from the second block on -- the first may still warm pools of the runtime)."""
    round_trip()
    marks = [free_now()]
    for _ in range(blocks):
        for _ in range(per_block):
            round_trip()
        marks.append(free_now())
    return marks, [(marks[i] - marks[i + 1]) / 2**20 for i in range(1, blocks)]


if __name__ == "__main__":
    marks, growth = measure()
    print("free device memory (MiB) after 1, 26, 51, 76, 101 round trips:", [round(m / 2**20, 1) for m in marks])
    print("lost per block of 25 round trips (MiB), blocks 2-4:", [round(x, 2) for x in growth])
    sys.exit(0 if max(growth) < 2.0 else 1)

#!/usr/bin/env python3
"""Does the allocation-free entry point run under HIP graph capture?  Captures one sdfgpu_nearest_dev_static call
(reordering on: curve keys + radix sort + walk) into a graph on a torch stream, replays it on fresh inputs and
compares with sdfgpu_nearest_dev."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdf_amd as S  # noqa: E402


def run(log=print, n=100_000, replays=3):
    """True when every replay on fresh inputs was bit-identical to sdfgpu_nearest_dev on the same inputs."""
    gpu = S.SdfGpu(0)
    gpu.upload_tree(S.random_scene_buffer(2000, 0x6A))
    L = S.default_extent(2000)
    rng = np.random.default_rng(0)

    def points():
        p = np.ones((n, 4), np.float32)
        p[:, :3] = rng.uniform(-L, L, (n, 3))
        return p

    pts = torch.from_numpy(points()).cuda()
    d = torch.empty(n, dtype=torch.float32, device="cuda")
    i = torch.empty(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):   # once outside capture: first-use set-up of the library and of hipcub
        gpu.nearest_dev_static(pts.data_ptr(), n, d.data_ptr(), i.data_ptr(), s.cuda_stream)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):   # capture mode "global": an allocation or a synchronisation inside would fail
        gpu.nearest_dev_static(pts.data_ptr(), n, d.data_ptr(), i.data_ptr(), torch.cuda.current_stream().cuda_stream)
    ok = True
    for rep in range(replays):
        pts.copy_(torch.from_numpy(points()).cuda())
        d.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        d2 = torch.empty_like(d)
        i2 = torch.empty_like(i)
        gpu.nearest_dev(pts.data_ptr(), n, d2.data_ptr(), i2.data_ptr(), 0)
        torch.cuda.synchronize()
        same = bool(torch.equal(d.view(torch.int32), d2.view(torch.int32)) and torch.equal(i, i2))
        log(f"replay {rep}: bit-identical to sdfgpu_nearest_dev: {same}")
        ok = ok and same
    return ok


if __name__ == "__main__":
    sys.exit(0 if run() else 1)

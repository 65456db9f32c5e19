#!/usr/bin/env python3
"""Query batch timings for one scene size (the environment decides SDFGPU_PREFETCH / SDFGPU_XCD_ORDER):
   tools/query_sweep.py <n_prims> [n_points]   ->  ms per batch, k = 1 and k = 8, unsorted and caller-sorted."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402

n_prims = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
buf = S.random_scene_buffer(n_prims, 0x5DF00005)
gpu = S.SdfGpu(0)
info = gpu.upload_tree(buf)
L = S.default_extent(n_prims)
rng = np.random.default_rng(5)
pts = np.ones((n, 4), np.float32)
pts[:, :3] = rng.uniform(-1.1 * L, 1.1 * L, (n, 3)).astype(np.float32)
far = float(os.environ.get("SWEEP_FAR_FRACTION", "0"))
if far > 0:   # a skewed batch: that share of the points 20-30 extents out in one direction (several times the cost per point)
    m = int(n * far)
    pts[:m, :3] = (np.array([25.0 * L, 0.0, 0.0]) + rng.uniform(-5 * L, 5 * L, (m, 3))).astype(np.float32)
dev = torch.device("cuda", 0)


def run(p, k, reorder):
    gpu.set_reorder(reorder)
    d = torch.empty(n * k, dtype=torch.float32, device=dev)
    i = torch.empty(n * k, dtype=torch.int32, device=dev)
    f = (lambda: gpu.nearest_dev(p.data_ptr(), n, d.data_ptr(), i.data_ptr(), 0)) if k == 1 else (lambda: gpu.knn_dev(p.data_ptr(), n, k, d.data_ptr(), i.data_ptr(), 0))
    f(); f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    gpu.set_reorder(True)
    return (time.perf_counter() - t0) * 100


t = torch.from_numpy(pts).to(dev)
res = {"unsorted k1": run(t, 1, True), "unsorted k8": run(t, 8, True)}
ts = torch.from_numpy(pts[np.argsort(W.hilbert_keys(pts[:, :3], float(np.abs(pts[:, :3]).max())), kind="stable")]).to(dev)
res["sorted k1"] = run(ts, 1, False)
res["sorted k8"] = run(ts, 8, False)
print(f"{n_prims} prims, tree {info.device_bytes / 2**20:.1f} MiB, prefetch env {os.environ.get('SDFGPU_PREFETCH', 'auto')}, xcd env {os.environ.get('SDFGPU_XCD_ORDER', 'default')}: "
      + ", ".join(f"{k} {v:.3f} ms" for k, v in res.items()))

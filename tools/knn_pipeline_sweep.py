"""Timing of sdfgpu_knn_dev / nearest_dev (config 5) against SDFGPU_OPT_QUERY_PIPELINE chunk counts."""
import sys, time
import torch
sys.path.insert(0, ".")
import sdf_amd as S
from sdf_amd import workloads as W, _ffi
wl = W.CONFIGS[5]
gpu = S.SdfGpu(0)
gpu.upload_tree(W.scene(wl))
n, k = wl.n_points, wl.k
pts = torch.from_numpy(W.query_points(wl, n)).cuda()
d = torch.empty(n * k, dtype=torch.float32, device="cuda"); i = torch.empty(n * k, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for chunks in (0, 2, 4, 8, 16):
    gpu.set_option(_ffi.OPT_QUERY_PIPELINE, chunks)
    for name, fn in (("knn8", lambda: gpu.knn_dev(pts.data_ptr(), n, k, d.data_ptr(), i.data_ptr(), st)),
                     ("k1", lambda: gpu.nearest_dev(pts.data_ptr(), n, d.data_ptr(), i.data_ptr(), st))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        print(f"chunks {chunks:2d} {name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)

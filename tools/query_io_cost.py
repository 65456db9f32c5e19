import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import sdf_amd as S
from sdf_amd import workloads as W
wl = W.CONFIGS[5]
gpu = S.SdfGpu(0); gpu.upload_tree(W.scene(wl))
dev = torch.device("cuda:0")
n = 10_000_000
for name, morton, reorder in (("unsorted, reorder on", False, True), ("presorted, reorder off", True, False), ("presorted, reorder on", True, True)):
    pts = torch.from_numpy(W.query_points(wl, n, morton=morton)).to(dev)
    d = torch.empty(n, dtype=torch.float32, device=dev); i = torch.empty(n, dtype=torch.int32, device=dev)
    d8 = torch.empty(n * 8, dtype=torch.float32, device=dev); i8 = torch.empty(n * 8, dtype=torch.int32, device=dev)
    gpu.set_reorder(reorder)
    st = torch.cuda.current_stream().cuda_stream
    for fn, label in ((lambda: gpu.nearest_dev(pts.data_ptr(), n, d.data_ptr(), i.data_ptr(), st), "k=1"),
                      (lambda: gpu.knn_dev(pts.data_ptr(), n, 8, d8.data_ptr(), i8.data_ptr(), st), "k=8")):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        print(f"{name:26s} {label}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")

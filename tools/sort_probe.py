"""Design probe: the k = 1 / k = 8 query on 10 M UNSORTED points, reordering on (whole call timed with events).
Under rocprofv3 --kernel-trace --stats it gives the per-kernel times of the batch sort for random input."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402

wl = W.CONFIGS[5]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
g = S.SdfGpu(0)
g.upload_tree(W.scene(wl))
pts = torch.from_numpy(W.query_points(wl, n)).cuda()
d = torch.empty(n * 8, dtype=torch.float32, device="cuda")
i = torch.empty(n * 8, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for k in (1, 8):
    f = (lambda: g.nearest_dev(pts.data_ptr(), n, d.data_ptr(), i.data_ptr(), st)) if k == 1 else (lambda: g.knn_dev(pts.data_ptr(), n, 8, d.data_ptr(), i.data_ptr(), st))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"k={k}: {e0.elapsed_time(e1) / 10:.4f} ms per batch of {n} unsorted points ({os.environ.get('SDFGPU_SORT', 'own sort')})")
g.close()

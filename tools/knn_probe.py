"""Design probe: the k = 8 walk alone -- 10 M caller-sorted points, library reordering off (no sort, no permutation)."""
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
g.set_reorder(False)
pts = torch.from_numpy(W.query_points(wl, n, morton=True)).cuda()
d = torch.empty(n * 8, dtype=torch.float32, device="cuda")
i = torch.empty(n * 8, dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    g.knn_dev(pts.data_ptr(), n, 8, d.data_ptr(), i.data_ptr(), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g.knn_dev(pts.data_ptr(), n, 8, d.data_ptr(), i.data_ptr(), st)
e1.record()
torch.cuda.synchronize()
print(f"k=8 caller-sorted, reorder off: {e0.elapsed_time(e1) / 10:.4f} ms per batch of {n} points")
g.close()

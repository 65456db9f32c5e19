"""Design probe (not product, not a test): the bench frame over the crate's literal buffer (PCA trans_inverse in every
merged box: chain path), isolated frame time.  python tools/crate_probe.py [config]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
wl = W.CONFIGS[cfg]
w, h = wl.width, wl.height
cam, mp = W.camera(wl, w, h), W.march(wl)
dev = torch.device("cuda:0")
buf = S.random_scene(wl.n_prims, wl.seed, opts={"pca_trans_inverse": True}).expanded().make_buffer()
g = S.SdfGpu(0)
info = g.upload_tree(buf)
o = (torch.empty(w * h, dtype=torch.uint8, device=dev), torch.empty(w * h, dtype=torch.float32, device=dev),
     torch.empty(w * h * 3, dtype=torch.float32, device=dev), torch.empty(w * h, dtype=torch.int32, device=dev))
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    g.render_dev(cam, w, h, mp, *(t.data_ptr() for t in o), st)
torch.cuda.synchronize()
ms = []
for _ in range(20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.render_dev(cam, w, h, mp, *(t.data_ptr() for t in o), st)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
print(f"walk {info.walk} levels {info.max_level} frame ms min {min(ms):.3f} avg {sum(ms) / len(ms):.3f} checksum {int(o[3].sum().item()) ^ int(o[0].sum().item())}")

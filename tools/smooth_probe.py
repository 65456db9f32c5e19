"""Design probe (not product, not a test): the general path's march on the headline scene with a smooth radius on every
union (SDFGPU_EXT_SMOOTH_UNION), frame time against the step cap.  python tools/smooth_probe.py [k] [div]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402

k = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
div = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wl = W.CONFIGS[3]
w, h = wl.width // div, wl.height // div
buf = W.scene(wl)
sm = S.SdfTreeBuffer(buf.downtree_buffer, buf.uptree_buffer.copy())
dt = sm.downtree_buffer
post = np.arange(len(dt), dtype=np.int64) + dt["len"].astype(np.int64) - (dt["level"].astype(np.int64) - 1)
un = (dt["is_primitive"] == 0) & (dt["op_code"] == 0)
sm.uptree_buffer["op_specific"]["floats"][post[un], 0] = k
cam = W.camera(wl, w, h)
dev = torch.device("cuda:0")
g = S.SdfGpu(0)
info = g.upload_tree(sm, extensions=S.EXT_SMOOTH_UNION)
o = (torch.empty(w * h, dtype=torch.uint8, device=dev), torch.empty(w * h, dtype=torch.float32, device=dev),
     torch.empty(w * h * 3, dtype=torch.float32, device=dev), torch.empty(w * h, dtype=torch.int32, device=dev))
st = torch.cuda.current_stream().cuda_stream
for steps in (1, 2, 4, 8, 16, 32, 64, 128):
    mp = W.march(wl)
    mp.max_steps = steps
    g.render_dev(cam, w, h, mp, *(t.data_ptr() for t in o), st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.render_dev(cam, w, h, mp, *(t.data_ptr() for t in o), st)
    torch.cuda.synchronize()
    print(f"walk {info.walk} k {k} {w}x{h} max_steps {steps}: {(time.perf_counter() - t0) * 1e3:.1f} ms, hits {int(o[0].sum().item())}")

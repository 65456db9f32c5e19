"""Rates of the HOST-pointer entry points (inputs/outputs cross PCIe inside the call).
Never the bench's `value`; recorded in DESIGN.md."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import sdf_amd as S
from sdf_amd import workloads as W

def t(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps

gpu = S.SdfGpu(0)
wl = W.CONFIGS[3]
gpu.upload_tree(W.scene(wl))
cam, mp = W.camera(wl), W.march(wl)
dt = t(lambda: gpu.render(cam, wl.width, wl.height, mp))
print(f"sdfgpu_render 1080p/10k, host outputs (43.5 MB): {dt*1e3:.2f} ms/frame = {wl.width*wl.height/dt/1e6:.1f} Mrays/s")
wl5 = W.CONFIGS[5]
gpu.upload_tree(W.scene(wl5))
pts = W.query_points(wl5)
dt = t(lambda: gpu.nearest(pts), 3)
print(f"sdfgpu_nearest 10M pts/100k prims, host in (160 MB) + out (80 MB): {dt*1e3:.1f} ms = {len(pts)/dt/1e6:.1f} Mqueries/s")
dt = t(lambda: gpu.knn(pts, 8), 3)
print(f"sdfgpu_knn k=8 10M pts, host in (160 MB) + out (640 MB): {dt*1e3:.1f} ms = {len(pts)/dt/1e6:.1f} Mqueries/s")

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sdf_amd as S
from oracle import oracle as O
buf = S.random_scene_buffer(300, 7)
pts = np.ones((16, 4), np.float32)
rng = np.random.default_rng(1)
pts[:, :3] = rng.uniform(-20, 20, (16, 3))
pts[1, 0] = np.nan
pts[2, 1] = np.inf
pts[3, 2] = -np.inf
pts[4, :3] = np.nan
pts[5, :3] = [3e38, -3e38, 3e38]
pts[6, :3] = [1e30, 0, 0]
pts[7, :3] = [np.inf, np.inf, np.inf]
for mode_name, mode in (("exh", O.MODE_EXHAUSTIVE), ("faux", O.MODE_FAUX)):
    d, i = O.nearest(buf, pts, mode=mode, threads=1)
    print("oracle", mode_name, d[:8], i[:8])
gpu = S.SdfGpu(0)
for b in (S.BOUNDS_BVH, S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE):
    gpu.upload_tree(buf, bounds=b)
    d, i = gpu.nearest(pts)
    print("gpu", b, d[:8], i[:8])
    dk, ik = gpu.knn(pts, 3)
    print("   knn", dk[1:8, 0], ik[1:8, 0])

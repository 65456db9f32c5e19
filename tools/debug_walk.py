"""GPU debugging aid: nearest() in every bounds mode against the oracle's exhaustive mode."""
import sys
import numpy as np
sys.path.insert(0, ".")
import sdf_amd as S
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
buf = S.random_scene_buffer(n, 7)
L = S.default_extent(n)
rng = np.random.default_rng(1)
pts = np.ones((4096, 4), np.float32)
pts[:, :3] = rng.uniform(-1.2 * L, 1.2 * L, (4096, 3)).astype(np.float32)
d0, i0 = O.nearest(buf, pts, mode=O.MODE_EXHAUSTIVE)
gpu = S.SdfGpu(0)
for name, mode in (("obb", S.BOUNDS_OBB), ("none", S.BOUNDS_NONE), ("aabb", S.BOUNDS_AABB), ("bvh", S.BOUNDS_BVH)):
    gpu.upload_tree(buf, bounds=mode)
    d, i = gpu.nearest(pts)
    bad = np.flatnonzero(d.view(np.uint32) != d0.view(np.uint32))
    print(name, "mismatch", len(bad), "of", len(pts), "idx mismatch", int((i != i0).sum()))
    for j in bad[:5]:
        print("   pt", j, "gpu", d[j], i[j], "oracle", d0[j], i0[j], "gpu>oracle" if d[j] > d0[j] else "gpu<oracle")

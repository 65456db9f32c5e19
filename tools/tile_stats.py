"""Diagnostic: per-tile work of the bench frame.  Picks tiles by their maximum step count
(from a CPU oracle run cached in /tmp/steps.npy if present) and prints the GPU counters of an
8x8 sub-frame render of exactly that tile."""
import sys
import numpy as np
sys.path.insert(0, ".")
import sdf_amd as S
from sdf_amd import workloads as W
from sdf_amd import _ffi

wl = W.CONFIGS[3]
w, h = wl.width, wl.height
gpu = S.SdfGpu(0)
gpu.upload_tree(W.scene(wl))
cam, mp = W.camera(wl), W.march(wl)
# full-frame steps from the GPU itself: one 8x8 render per tile is too slow, so sample tiles
rng = np.random.default_rng(0)
tx, ty = w // 8, h // 8
rows = []
for t in rng.choice(tx * ty, 600, replace=False):
    x0, y0 = (t % tx) * 8, (t // tx) * 8
    c = _ffi.Camera()
    for a in range(3):
        c.eye[a] = cam.eye[a]
        c.du[a] = cam.du[a]
        c.dv[a] = cam.dv[a]
        c.origin[a] = np.float32(np.float32(cam.origin[a] + np.float32(cam.du[a] * np.float32(x0))) + np.float32(cam.dv[a] * np.float32(y0)))
    st = gpu.render_stats(c, 8, 8, mp)
    rows.append((st["wave_steps"], st["wave_walks"], st["wave_inner"], st["wave_leaf"], st["steps"], st["queries"], int(t)))
rows.sort(reverse=True)
print("wave_steps wave_walks wave_inner wave_leaf lane_steps lane_walks tile")
for r in rows[:12]:
    print(*r, " visits/walk %.1f" % (r[2] / max(1, r[1])))
print("...")
for r in rows[300:304]:
    print(*r, " visits/walk %.1f" % (r[2] / max(1, r[1])))
tot = np.array([r[:6] for r in rows], dtype=np.float64)
print("sample mean: wave_steps %.1f wave_inner %.0f wave_leaf %.0f" % (tot[:, 0].mean(), tot[:, 2].mean(), tot[:, 3].mean()))
long_ = tot[tot[:, 0] >= 128]
print("tiles with 128 steps: %d of %d; their mean wave_inner %.0f (%.1f per walk), lane walks per wave-walk %.2f" % (
    len(long_), len(tot), long_[:, 2].mean(), (long_[:, 2] / long_[:, 1]).mean(), (long_[:, 5] / long_[:, 1]).mean()))

import os, sys
os.environ["SDFGPU_PASS_STEPS"] = "12,20,28,40,56,80"
import numpy as np
sys.path.insert(0, ".")
import sdf_amd as S
from oracle import oracle as O
g = S.SdfGpu(0)
n, w, h = 250, 121, 67
buf = S.random_scene_buffer(n, 0xBEEF)
L = S.default_extent(n)
cam = S.camera_look_at((0.2 * L, 0.1 * L, -1.2 * L), (0, 0, 0), (0, 1, 0), 1.0, w, h)
for ms in (28, 57):
    mp = S.march_params(max_steps=ms, eps_hit=1e-3, t_min=0.0, t_max=6 * L, normal_h=1e-4)
    ref = O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, threads=8)
    g.upload_tree(buf)
    for on in (True, False):
        g.set_compaction(on)
        img = g.render(cam, w, h, mp)
        bad = np.flatnonzero(img["hit"] != ref["hit"])
        badp = np.flatnonzero(img["prim"] != ref["prim"])
        print("max_steps", ms, "compaction", on, "hit mismatches", len(bad), "prim mismatches", len(badp))
        for i in bad[:10]:
            print("  pix", i % w, i // w, "gpu hit", img["hit"][i], "ref hit", ref["hit"][i], "ref steps", ref["steps"][i], "gpu depth", img["depth"][i], "ref depth", ref["depth"][i])

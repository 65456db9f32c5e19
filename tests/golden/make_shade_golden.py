#!/usr/bin/env python3
"""Generates tests/golden/shade_300.ppm: the oracle's render of a 300-primitive scene (160 x 96), shaded
the way the crate's legacy fragment shader shades (src/main__.rs:163-169; tools/render_ppm.py::shade).
Regression image for SURVEY 8f N-4.  Run from the repo root:  python tests/golden/make_shade_golden.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from render_ppm import shade  # noqa: E402
from test_gpu_shade import scene_and_camera, W_, H_  # noqa: E402

buf, cam, mp = scene_and_camera()
img = O.render(buf, cam, W_, H_, mp, mode=O.MODE_TUNED, threads=8)
rgb = shade(img, cam, W_, H_)
with open(os.path.join(ROOT, "tests", "golden", "shade_300.ppm"), "wb") as f:
    f.write(b"P6\n%d %d\n255\n" % (W_, H_))
    f.write(rgb.tobytes())
print("hit pixels", int(img["hit"].sum()))

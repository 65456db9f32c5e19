#!/usr/bin/env python3
"""Render a BASELINE scene on the GPU and write a PPM, shaded as the crate's legacy fragment
shader does (src/main__.rs:163-169: intensity = max(0, dot(normalize(-p), n)); one colour per
primitive).  Visual regression aid; not on the measured path.

usage: tools/render_ppm.py [--config 3] [--width W --height H] out.ppm
"""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")
import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402


def shade(img, cam, w, h):
    hit = img["hit"].astype(bool)
    eye = np.array(cam.eye, np.float32)
    origin, du, dv = (np.array(v, np.float32) for v in (cam.origin, cam.du, cam.dv))
    idx = np.arange(w * h)
    x = (idx % w).astype(np.float32)[:, None]
    y = (idx // w).astype(np.float32)[:, None]
    d = ((origin + du * x) + dv * y) - eye
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = eye + d * np.where(hit, img["depth"], 0.0)[:, None]
    to_origin = -p / np.maximum(np.linalg.norm(p, axis=1, keepdims=True), 1e-20)
    inten = np.maximum(0.0, np.sum(to_origin * img["normal"], axis=1))
    prim = img["prim"].astype(np.uint64)
    # a fixed colour per primitive id (the crate stores one per sphere; here: hashed)
    hsh = (prim * np.uint64(2654435761)) & np.uint64(0xFFFFFF)
    col = np.stack([(hsh >> np.uint64(16)) & np.uint64(255), (hsh >> np.uint64(8)) & np.uint64(255), hsh & np.uint64(255)], 1)
    col = 0.35 + 0.65 * col.astype(np.float32) / 255.0
    rgb = np.where(hit[:, None], col * (0.15 + 0.85 * inten[:, None]), np.array([[0.05, 0.06, 0.09]], np.float32))
    return (np.clip(rgb, 0, 1) * 255).astype(np.uint8).reshape(h, w, 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("out")
    a = ap.parse_args()
    wl = W.CONFIGS[a.config]
    w, h = a.width or wl.width or 1920, a.height or wl.height or 1080
    gpu = S.SdfGpu(0)
    gpu.upload_tree(W.scene(wl))
    cam, mp = W.camera(wl, w, h), W.march(wl)
    img = gpu.render(cam, w, h, mp)
    rgb = shade(img, cam, w, h)
    with open(a.out, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(rgb.tobytes())
    print(f"{a.out}: {w}x{h}, {int(img['hit'].sum())} hit pixels")


if __name__ == "__main__":
    main()

"""Shading / image writer: a frame shaded as the crate's legacy fragment shader shades
(src/main__.rs:163-169: Lambert against the direction to the origin, one colour per primitive;
tools/render_ppm.py) against a committed regression image (tests/golden/shade_300.ppm, made from the
ORACLE's render by tests/golden/make_shade_golden.py)."""
import math
import os
import sys

import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from render_ppm import shade  # noqa: E402

W_, H_ = 160, 96
GOLDEN = os.path.join(ROOT, "tests", "golden", "shade_300.ppm")


def scene_and_camera():
    n = 300
    buf = S.random_scene_buffer(n, 0x5DF0C010)
    L = S.default_extent(n)
    cam = S.camera_look_at((0.4 * L, 0.3 * L, -2.2 * L), (0, 0, 0), (0, 1, 0), math.pi / 4, W_, H_)
    mp = S.march_params(max_steps=128, eps_hit=1e-3, t_min=0.0, t_max=6 * L, normal_h=1e-3)
    return buf, cam, mp


def read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = (int(v) for v in f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)


def _check(rgb):
    want = read_ppm(GOLDEN)
    assert want.shape == rgb.shape
    # one grey level of slack: the shading is host numpy arithmetic, whose last bit may differ between hosts
    assert int(np.abs(rgb.astype(np.int16) - want.astype(np.int16)).max()) <= 1
    assert (want.reshape(-1, 3) != want[0, 0]).any()      # not an empty frame


def test_oracle_frame_shades_to_the_regression_image():
    buf, cam, mp = scene_and_camera()
    _check(shade(O.render(buf, cam, W_, H_, mp, mode=O.MODE_TUNED, threads=4), cam, W_, H_))


@pytest.mark.gpu
def test_gpu_frame_shades_to_the_regression_image(gpu):
    buf, cam, mp = scene_and_camera()
    gpu.upload_tree(buf)
    _check(shade(gpu.render(cam, W_, H_, mp), cam, W_, H_))

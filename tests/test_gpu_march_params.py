"""March parameters at their edges (step cap 0 / 1, eps 0 / negative / huge, far plane before the near one,
negative near plane, normal step 0 / huge) and outside the march's domain (eye or t_min beyond 2^60 or not
numbers, a pixel without a direction), GPU against the oracle's march, bit for bit: the loop contract
(oracle.cpp oracle_render) holds for any parameter values, not only sensible ones."""
import math

import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits

pytestmark = pytest.mark.gpu

W, H = 88, 56


def _same(img, ref):
    assert np.array_equal(img["hit"], ref["hit"])
    assert np.array_equal(img["prim"], ref["prim"])
    assert np.array_equal(bits(img["depth"]), bits(ref["depth"]))
    assert np.array_equal(bits(img["normal"]), bits(ref["normal"]))


CASES = {
    "no steps": dict(max_steps=0),
    "one step": dict(max_steps=1),
    "two steps": dict(max_steps=2),
    "eps zero": dict(eps_hit=0.0),
    "eps negative": dict(eps_hit=-1.0),
    "eps huge": dict(eps_hit=1e9),
    "far plane at zero": dict(t_max=0.0),
    "far plane before the near plane": dict(t_min=40.0, t_max=10.0),
    "near plane behind the eye": dict(t_min=-30.0),
    "near plane inside the scene": dict(t_min=120.0),
    "normal step zero": dict(normal_h=0.0),
    "normal step huge": dict(normal_h=1e6),
    "normal step negative": dict(normal_h=-1e-3),
    "far plane infinite": dict(t_max=float("inf"), max_steps=40),
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("bounds", [S.BOUNDS_BVH, S.BOUNDS_OBB])
def test_march_parameter_edges(gpu, name, bounds):
    n = 200
    L = S.default_extent(n)
    buf = S.random_scene_buffer(n, 0xED6E)
    cam = S.camera_look_at((0.1 * L, 0.05 * L, -2.2 * L), (0, 0, 0), (0, 1, 0), math.pi / 4, W, H)
    kw = dict(max_steps=64, eps_hit=1e-3, t_min=0.0, t_max=6.0 * L, normal_h=1e-4)
    kw.update(CASES[name])
    mp = S.march_params(**kw)
    ref = O.render(buf, cam, W, H, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    gpu.upload_tree(buf, bounds=bounds)
    _same(gpu.render(cam, W, H, mp), ref)


@pytest.mark.parametrize("what", ["eye nan", "eye inf", "eye 1e30", "eye 2e18", "t_min nan", "t_min 1e19", "t_max nan"])
def test_camera_outside_the_march_domain(gpu, what):
    # Eye coordinates / t_min beyond 2^60 (1.15e18) or not numbers: every ray is a miss outright, on the GPU
    # and in the oracle's march, whatever the nearest-primitive arithmetic would have made of such points.
    # A NaN far plane is no far plane (rays end at 2^60 or at the step cap).
    n = 50
    L = S.default_extent(n)
    buf = S.random_scene_buffer(n, 0xED6F)
    cam = S.camera_look_at((0.0, 0.0, -2.5 * L), (0, 0, 0), (0, 1, 0), 0.5, W, H)
    kw = dict(max_steps=16, eps_hit=1e-3, t_min=0.0, t_max=6.0 * L, normal_h=1e-4)
    if what.startswith("eye"):
        cam.eye[0] = {"nan": float("nan"), "inf": float("inf"), "1e30": 1e30, "2e18": -2e18}[what.split()[1]]
    elif what == "t_min nan":
        kw["t_min"] = float("nan")
    elif what == "t_min 1e19":
        kw["t_min"] = 1e19
    else:
        kw["t_max"] = float("nan")
    mp = S.march_params(**kw)
    ref = O.render(buf, cam, W, H, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    if what != "t_max nan":
        assert ref["hit"].sum() == 0 and np.all(np.isposinf(ref["depth"])) and np.all(ref["prim"] == S.NO_PRIM)
    else:
        assert ref["hit"].sum() > 0
    for bounds in (S.BOUNDS_BVH, S.BOUNDS_OBB, S.BOUNDS_NONE):
        gpu.upload_tree(buf, bounds=bounds)
        _same(gpu.render(cam, W, H, mp), ref)


def test_degenerate_pixel_is_a_miss(gpu):
    # the image plane's origin put on the eye: pixel (0, 0) has the zero vector for a direction (0 / 0)
    import copy
    n = 50
    L = S.default_extent(n)
    buf = S.random_scene_buffer(n, 0xED70)
    cam = S.camera_look_at((0.0, 0.0, -2.5 * L), (0, 0, 0), (0, 1, 0), 0.5, W, H)
    bad = copy.copy(cam)
    for a in range(3):
        bad.origin[a] = bad.eye[a]
    mp = S.march_params(max_steps=32, eps_hit=1e-3, t_min=0.0, t_max=6.0 * L, normal_h=1e-4)
    ref = O.render(buf, bad, W, H, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    assert ref["hit"][0] == 0 and np.isposinf(ref["depth"][0])
    gpu.upload_tree(buf)
    _same(gpu.render(bad, W, H, mp), ref)


def test_general_path_camera_outside_the_march_domain(gpu):
    buf = S.random_scene(30, 0xED71, opts={"pca_trans_inverse": True}).expanded().make_buffer()
    assert gpu.upload_tree(buf).identity_inner == 0
    cam = S.camera_look_at((0.0, 0.0, -40.0), (0, 0, 0), (0, 1, 0), 0.5, W, H)
    mp = S.march_params(max_steps=16, eps_hit=1e-3, t_min=0.0, t_max=float("inf"), normal_h=1e-4)
    _same(gpu.render(cam, W, H, mp), O.render(buf, cam, W, H, mp, mode=O.MODE_EXHAUSTIVE, threads=8))
    cam.eye[1] = float("nan")
    img = gpu.render(cam, W, H, mp)
    _same(img, O.render(buf, cam, W, H, mp, mode=O.MODE_EXHAUSTIVE, threads=8))
    assert img["hit"].sum() == 0

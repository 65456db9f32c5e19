"""Contexts are independent: several host threads, each with a context of its own on
the same GPU, upload, query and render at the same time (ctypes drops the GIL for the duration of a call) and
each gets exactly what a single-threaded run gets."""
import threading

import numpy as np
import pytest

import sdf_amd as S
from conftest import bits, make_points

pytestmark = pytest.mark.gpu


def test_four_threads_four_contexts(gpu):
    scenes = [S.random_scene_buffer(200 + 50 * t, 0x7EAD + t) for t in range(4)]
    pts = [make_points(6000, S.default_extent(300), seed=t) for t in range(4)]
    cam = S.camera_look_at((0, 0, -90.0), (0, 0, 0), (0, 1, 0), 0.8, 160, 96)
    mp = S.march_params(max_steps=48, eps_hit=1e-3, t_min=0.0, t_max=400.0, normal_h=1e-4)
    want = []
    for t in range(4):
        gpu.upload_tree(scenes[t])
        want.append((gpu.nearest(pts[t]), gpu.knn(pts[t][:1500], 8), gpu.render(cam, 160, 96, mp)))
    errors = []

    def work(t):
        try:
            g = S.SdfGpu(0)
            for _ in range(6):
                g.upload_tree(scenes[t])
                d, i = g.nearest(pts[t])
                dk, ik = g.knn(pts[t][:1500], 8)
                img = g.render(cam, 160, 96, mp)
                (d0, i0), (dk0, ik0), img0 = want[t]
                ok = (np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0) and np.array_equal(bits(dk), bits(dk0))
                      and np.array_equal(ik, ik0) and all(np.array_equal(np.asarray(img[q]).view(np.uint8), np.asarray(img0[q]).view(np.uint8))
                                                         for q in ("hit", "depth", "normal", "prim")))
                if not ok:
                    errors.append(f"thread {t}: results differ")
            g.close()
        except Exception as e:   # noqa: BLE001
            errors.append(f"thread {t}: {e!r}")

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=100)
    assert not any(th.is_alive() for th in threads), "a thread is stuck"
    assert not errors, errors

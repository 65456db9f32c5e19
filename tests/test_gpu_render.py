"""GPU parity: sphere trace + normals (K3/K4) against oracle_render.

Hit masks and primitive indices bit-exact; depths and normals bit-exact too
(the north star asks 1e-5 relative; the kernels reproduce the op order).
"""
import math

import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits

pytestmark = pytest.mark.gpu


def scene_camera(n, w, h):
    L = S.default_extent(n)
    cam = S.camera_look_at((0, 0, -2.5 * L), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
    mp = S.march_params(max_steps=128, eps_hit=1e-3, t_min=0.0, t_max=6 * L, normal_h=1e-4)
    return cam, mp


def _compare(img, ref):
    assert np.array_equal(img["hit"], ref["hit"])
    assert np.array_equal(img["prim"], ref["prim"])
    assert np.array_equal(bits(img["depth"]), bits(ref["depth"]))
    assert np.array_equal(bits(img["normal"]), bits(ref["normal"]))


@pytest.mark.parametrize("bounds", [S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_BVH])
@pytest.mark.parametrize("n,w,h", [(1, 64, 48), (7, 64, 48), (64, 96, 64), (1000, 160, 96)])
def test_render_matches_oracle(gpu, n, w, h, bounds):
    buf = S.random_scene_buffer(n, 0x5DF00100 + n)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf, bounds=bounds)
    img = gpu.render(cam, w, h, mp)
    ref = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE if n <= 64 else O.MODE_TUNED, threads=8)
    assert ref["hit"].sum() > 0
    _compare(img, ref)


def test_render_ragged_image(gpu):
    # width / height not multiples of the 8x8 tile
    n, w, h = 40, 61, 37
    buf = S.random_scene_buffer(n, 123)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf)
    _compare(gpu.render(cam, w, h, mp), O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8))


def test_render_empty_scene_all_miss(gpu):
    gpu.upload_tree(S.SdfTreeBuffer.make_empty())
    cam, mp = scene_camera(10, 32, 16)
    img = gpu.render(cam, 32, 16, mp)
    assert img["hit"].sum() == 0 and np.all(np.isposinf(img["depth"])) and np.all(img["prim"] == S.NO_PRIM)
    assert np.all(img["normal"] == 0)
    # no far plane either: every ray marches all its steps through an empty scene and still ends as a miss
    mp2 = S.march_params(max_steps=16, eps_hit=1e-3, t_min=0.0, t_max=float("inf"), normal_h=1e-4)
    img = gpu.render(cam, 32, 16, mp2)
    assert img["hit"].sum() == 0 and np.all(img["prim"] == S.NO_PRIM)


def test_sphere_normal_points_outward(gpu):
    # SURVEY E-7: sphere at c, hit point p -> normal ~ (p - c)/|p - c|
    c = np.array([0.5, -0.25, 2.0], np.float32)
    buf = (S.SdfBuilder.primitive(S.SdfSphere(1.0)).transform(S.Transform.from_translation(c))
           .finalize().expanded().make_buffer())
    w = h = 64
    cam = S.camera_look_at((0, 0, -6), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
    mp = S.march_params(max_steps=64, eps_hit=1e-4, t_min=0.0, t_max=50.0, normal_h=1e-3)
    gpu.upload_tree(buf)
    img = gpu.render(cam, w, h, mp)
    _compare(img, O.render(buf, cam, w, h, mp, mode=O.MODE_FAUX))
    hit = img["hit"].astype(bool)
    assert hit.sum() > 100
    # reconstruct hit points from depth and check the analytic normal
    ys, xs = np.divmod(np.flatnonzero(hit), w)
    org = np.array(cam.origin) + np.outer(xs, np.array(cam.du)) + np.outer(ys, np.array(cam.dv))
    d = org - np.array(cam.eye)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    p = np.array(cam.eye) + d * img["depth"][hit][:, None]
    nrm = (p - c) / np.linalg.norm(p - c, axis=1, keepdims=True)
    assert np.abs(img["normal"][hit] - nrm).max() < 2e-3


def test_tiles_and_untile_roundtrip(gpu):
    torch = pytest.importorskip("torch")
    n, w, h = 100, 100, 52
    buf = S.random_scene_buffer(n, 5)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf)
    full = gpu.render(cam, w, h, mp)
    tiles = ((w + 7) // 8) * ((h + 7) // 8)
    dev = torch.device("cuda:0")
    o_hit = torch.zeros(w * h, dtype=torch.uint8, device=dev)
    o_depth = torch.zeros(w * h, dtype=torch.float32, device=dev)
    o_normal = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    o_prim = torch.zeros(w * h, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for parts in (1, 3):
        for r in range(parts):
            cnt = len(range(r, tiles, parts))
            t_hit = torch.empty(cnt * 64, dtype=torch.uint8, device=dev)
            t_depth = torch.empty(cnt * 64, dtype=torch.float32, device=dev)
            t_normal = torch.empty(cnt * 64 * 3, dtype=torch.float32, device=dev)
            t_prim = torch.empty(cnt * 64, dtype=torch.int32, device=dev)
            gpu.render_tiles_dev(cam, w, h, mp, r, parts, cnt, t_hit.data_ptr(), t_depth.data_ptr(),
                                 t_normal.data_ptr(), t_prim.data_ptr(), stream)
            gpu.untile_dev(w, h, r, parts, cnt, t_hit.data_ptr(), t_depth.data_ptr(), t_normal.data_ptr(),
                           t_prim.data_ptr(), o_hit.data_ptr(), o_depth.data_ptr(), o_normal.data_ptr(),
                           o_prim.data_ptr(), stream)
        torch.cuda.synchronize()
        assert np.array_equal(o_hit.cpu().numpy(), full["hit"])
        assert np.array_equal(bits(o_depth.cpu().numpy()), bits(full["depth"]))
        assert np.array_equal(bits(o_normal.cpu().numpy().reshape(-1, 3)), bits(full["normal"]))
        assert np.array_equal(o_prim.cpu().numpy().view(np.uint32), full["prim"])


@pytest.mark.parametrize("ext", [0.0, -1.0, 0.05, 3.0, 50.0])
def test_walk_skipping_never_changes_the_frame(gpu, ext):
    # SDFGPU_OPT_SKIP_EXTENSION only trades walks for proofs; every output stays bit-identical
    n, w, h = 300, 128, 80
    buf = S.random_scene_buffer(n, 0x5C1F)
    cam, mp = scene_camera(n, w, h)
    ref = O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, threads=8)
    for bounds in (S.BOUNDS_AABB, S.BOUNDS_OBB):
        gpu.upload_tree(buf, bounds=bounds)
        gpu.set_skip_extension(ext)
        _compare(gpu.render(cam, w, h, mp), ref)
    gpu.set_skip_extension(-1.0)


def test_grazing_camera_inside_the_scene(gpu):
    # camera inside the primitive cloud, long grazing rays: the regime walk skipping is for
    n, w, h = 400, 96, 64
    buf = S.random_scene_buffer(n, 0xC0FFEE)
    L = S.default_extent(n)
    cam = S.camera_look_at((0.3 * L, 0.1 * L, -0.2 * L), (L, 0.2 * L, 0.9 * L), (0, 1, 0), 1.2, w, h)
    mp = S.march_params(max_steps=200, eps_hit=1e-4, t_min=0.0, t_max=6 * L, normal_h=1e-3)
    gpu.upload_tree(buf)
    st = gpu.render_stats(cam, w, h, mp)
    assert st["steps"] > st["queries"] > 0          # some steps were proven without a walk
    _compare(gpu.render(cam, w, h, mp), O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, threads=8))


@pytest.mark.parametrize("max_steps", [0, 1, 11, 12, 13, 28, 57, 200])
def test_ray_compaction_is_invisible(max_steps, monkeypatch):
    # the multi-pass march (SDFGPU_OPT_COMPACTION) re-packs survivors between passes; pass
    # boundaries -- here a six-pass schedule, read when the context is created -- must not show
    # in any output
    monkeypatch.setenv("SDFGPU_PASS_STEPS", "12,20,28,40,56,80")
    g = S.SdfGpu(0)
    try:
        n, w, h = 250, 121, 67
        buf = S.random_scene_buffer(n, 0xBEEF)
        L = S.default_extent(n)
        cam = S.camera_look_at((0.2 * L, 0.1 * L, -1.2 * L), (0, 0, 0), (0, 1, 0), 1.0, w, h)
        mp = S.march_params(max_steps=max_steps, eps_hit=1e-3, t_min=0.0, t_max=6 * L, normal_h=1e-4)
        ref = O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, threads=8)
        g.upload_tree(buf)
        for on in (True, False):
            g.set_compaction(on)
            _compare(g.render(cam, w, h, mp), ref)
    finally:
        g.close()


def test_compaction_two_frames_in_flight(gpu):
    # two frames on two streams of one context, compaction on: each stream owns its
    # pass state, so both frames equal the single-stream frame bit for bit
    import torch

    n, w, h = 400, 320, 200
    buf = S.random_scene_buffer(n, 0xF00D)
    L = S.default_extent(n)
    cams = [
        S.camera_look_at((0.2 * L, 0.1 * L, -1.2 * L), (0, 0, 0), (0, 1, 0), 1.0, w, h),
        S.camera_look_at((-0.9 * L, 0.4 * L, 0.8 * L), (0, 0, 0), (0, 1, 0), 0.8, w, h),
    ]
    mp = S.march_params(max_steps=96, eps_hit=1e-3, t_min=0.0, t_max=6 * L, normal_h=1e-4)
    gpu.upload_tree(buf)
    gpu.set_compaction(False)
    want = [gpu.render(c, w, h, mp) for c in cams]
    gpu.set_compaction(True)
    try:
        dev = torch.device("cuda:0")
        streams = [torch.cuda.Stream(dev) for _ in cams]
        outs = []
        for _ in cams:
            outs.append((
                torch.empty(w * h, dtype=torch.uint8, device=dev),
                torch.empty(w * h, dtype=torch.float32, device=dev),
                torch.empty(w * h * 3, dtype=torch.float32, device=dev),
                torch.empty(w * h, dtype=torch.int32, device=dev),
            ))
        torch.cuda.synchronize()
        for rep in range(3):
            for cam, st, o in zip(cams, streams, outs):
                gpu.render_dev(cam, w, h, mp, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                               stream=st.cuda_stream)
        torch.cuda.synchronize()
        for o, ref in zip(outs, want):
            got = {
                "hit": o[0].cpu().numpy(),
                "depth": o[1].cpu().numpy(),
                "normal": o[2].cpu().numpy().reshape(-1, 3),
                "prim": o[3].cpu().numpy().view(np.uint32),
            }
            _compare(got, ref)
    finally:
        gpu.set_compaction(False)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_packed_planes_assembly_as_in_bench(gpu, world):
    # the multi-GPU data path of bench.py on one GPU: every rank's tiles into its packed
    # [depth | normal | prim | hit] buffer, the buffers laid end to end as the all-gather
    # leaves them, untile from each chunk -> the whole-frame render
    import torch
    from sdf_amd import partition as P

    n, w, h = 200, 173, 99          # ragged: partial tiles on both edges, tile count not a multiple of world
    buf = S.random_scene_buffer(n, 77)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf)
    full = gpu.render(cam, w, h, mp)
    tiles = P.n_tiles(w, h)
    nbytes = P.packed_bytes(world, tiles)
    dev = torch.device("cuda:0")
    gathered = torch.zeros(world * nbytes, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for r in range(world):
        mine = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        first, stride, cnt = P.rank_tiles(r, world, tiles)
        ptrs = tuple(t.data_ptr() for t in P.planes(mine, world, tiles))
        gpu.render_tiles_dev(cam, w, h, mp, first, stride, cnt, *ptrs, stream)
        gathered[r * nbytes:(r + 1) * nbytes] = mine
    o_hit = torch.zeros(w * h, dtype=torch.uint8, device=dev)
    o_depth = torch.zeros(w * h, dtype=torch.float32, device=dev)
    o_normal = torch.zeros(w * h * 3, dtype=torch.float32, device=dev)
    o_prim = torch.zeros(w * h, dtype=torch.int32, device=dev)
    for r in range(world):
        part = P.planes(gathered[r * nbytes:(r + 1) * nbytes], world, tiles)
        rf, rs, rc = P.rank_tiles(r, world, tiles)
        gpu.untile_dev(w, h, rf, rs, rc, *(t.data_ptr() for t in part), o_hit.data_ptr(), o_depth.data_ptr(),
                       o_normal.data_ptr(), o_prim.data_ptr(), stream)
    torch.cuda.synchronize()
    got = {"hit": o_hit.cpu().numpy(), "depth": o_depth.cpu().numpy(),
           "normal": o_normal.cpu().numpy().reshape(-1, 3), "prim": o_prim.cpu().numpy().view(np.uint32)}
    _compare(got, full)

    # the same assembly in ONE launch (sdfgpu_untile_gathered_dev, what sdfgpu_render_frame_comm's root runs):
    # rank `me`'s share is read from its own buffer, not from the gathered one (poisoned here to prove it)
    assert S._ffi.load_library().sdfgpu_packed_share_bytes(w, h, world) == nbytes
    me = world - 1
    mine = gathered[me * nbytes:(me + 1) * nbytes].clone()
    gathered[me * nbytes:(me + 1) * nbytes] = 0xA5
    o2 = [torch.zeros_like(t) for t in (o_hit, o_depth, o_normal, o_prim)]
    gpu.untile_gathered_dev(w, h, world, me, gathered.data_ptr(), mine.data_ptr(), *(t.data_ptr() for t in o2), stream)
    torch.cuda.synchronize()
    got = {"hit": o2[0].cpu().numpy(), "depth": o2[1].cpu().numpy(),
           "normal": o2[2].cpu().numpy().reshape(-1, 3), "prim": o2[3].cpu().numpy().view(np.uint32)}
    _compare(got, full)


def _scale_leaf(buf, i, s):
    """Scale the whole world->primitive map of down-block i by s (a non-rigid leaf: its value is
    s x the distance a rigid copy would report, minus the radius)."""
    m = buf.downtree_buffer["bounding_box"]["trans_inverse"][i]
    m[[0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14]] *= np.float32(s)


@pytest.mark.parametrize("bounds", [S.BOUNDS_AUTO, S.BOUNDS_NONE])
def test_render_non_rigid_leaves_match_exhaustive(gpu, bounds):
    # the march's walk skipping assumed a leaf value moves at most delta when the point
    # moves delta; a scaled leaf moves faster.  The budget now decays by the leaves' measured
    # Lipschitz constant, so NONE mode (the only one that accepts such leaves) matches the oracle.
    n, w, h = 120, 96, 64
    buf = S.random_scene_buffer(n, 0x5DF00777)
    prim_rows = np.flatnonzero(buf.downtree_buffer["is_primitive"])
    rng = np.random.default_rng(3)
    for i in prim_rows[rng.random(len(prim_rows)) < 0.5]:
        _scale_leaf(buf, int(i), rng.choice([0.5, 2.0, 3.0]))
    cam, mp = scene_camera(n, w, h)
    info = gpu.upload_tree(buf, bounds=bounds)
    assert info.bounds == S.BOUNDS_NONE and info.rigid_leaves == 0
    ref = O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE, threads=8)
    assert ref["hit"].sum() > 0
    _compare(gpu.render(cam, w, h, mp), ref)
    with pytest.raises(S.SdfGpuError):
        gpu.upload_tree(buf, bounds=S.BOUNDS_OBB)


def test_render_scaled_leaf_counter_example(gpu):
    # the advisor's hand-worked case: ray from the origin along +z, rigid sphere A (r = 1) at
    # (0, 60, 100), sphere B (r = 1) at (0, -3, 150) whose trans_inverse is scaled by 2
    a = S.SdfBuilder.primitive(S.SdfSphere(1.0)).transform(S.Transform.from_xyz(0, 60, 100))
    b = S.SdfBuilder.primitive(S.SdfSphere(1.0)).transform(S.Transform.from_xyz(0, -3, 150))
    buf = a.operation(S.SdfUnion(0.0)).with_(b).finalize().expanded().make_buffer()
    rows = np.flatnonzero(buf.downtree_buffer["is_primitive"])
    _scale_leaf(buf, int(rows[1]), 2.0)
    w = h = 16
    cam = S.camera_look_at((0, 0, 0), (0, 0, 1), (0, 1, 0), 0.02, w, h)
    mp = S.march_params(max_steps=64, eps_hit=1e-3, t_min=0.0, t_max=400.0, normal_h=1e-3)
    gpu.upload_tree(buf)
    _compare(gpu.render(cam, w, h, mp), O.render(buf, cam, w, h, mp, mode=O.MODE_EXHAUSTIVE))


def test_more_streams_than_scratch_slots(gpu):
    # per-stream march scratch is recycled least-recently-used beyond 16 streams (it used to fail with
    # SDFGPU_ERR_TOO_LARGE once 16 distinct streams had rendered with compaction on)
    import torch

    n, w, h = 150, 96, 64
    buf = S.random_scene_buffer(n, 0x57E4)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf)
    want = gpu.render(cam, w, h, mp)
    gpu.set_compaction(True)
    try:
        dev = torch.device("cuda:0")
        for k in range(20):
            st = torch.cuda.Stream(dev)
            o = (torch.zeros(w * h, dtype=torch.uint8, device=dev), torch.zeros(w * h, dtype=torch.float32, device=dev),
                 torch.zeros(w * h * 3, dtype=torch.float32, device=dev), torch.zeros(w * h, dtype=torch.int32, device=dev))
            gpu.render_dev(cam, w, h, mp, *(t.data_ptr() for t in o), st.cuda_stream)
            st.synchronize()
            assert np.array_equal(o[0].cpu().numpy(), want["hit"]), f"stream {k}"
            assert np.array_equal(bits(o[1].cpu().numpy()), bits(want["depth"])), f"stream {k}"
    finally:
        gpu.set_compaction(False)


def test_first_march_step_shared_or_not(gpu):
    # With t_min = 0 every ray's first step stands at the eye and the walk from there is made once per frame
    # (kernels.hip, eye_walk_kernel); with any other t_min, or for a ray whose direction is degenerate, the
    # wavefronts walk for themselves.  All of it must render what the oracle renders.
    n, w, h = 300, 96, 64
    buf = S.random_scene_buffer(n, 0xE7E)
    L = S.default_extent(n)
    gpu.upload_tree(buf)
    cam = S.camera_look_at((0, 0, -2.5 * L), (0, 0, 0), (0, 1, 0), math.pi / 4, w, h)
    for t_min in (0.0, 0.25 * L, -3.0):
        mp = S.march_params(max_steps=96, eps_hit=1e-3, t_min=t_min, t_max=6 * L, normal_h=1e-4)
        _compare(gpu.render(cam, w, h, mp), O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, threads=8))
    # the eye inside the scene, next to (and for some seeds inside) primitives: the shared first step can hit
    cam_in = S.camera_look_at((0.3, -0.2, 0.1), (0, 0, 5.0), (0, 1, 0), math.pi / 3, w, h)
    mp = S.march_params(max_steps=96, eps_hit=1e-3, t_min=0.0, t_max=6 * L, normal_h=1e-4)
    _compare(gpu.render(cam_in, w, h, mp), O.render(buf, cam_in, w, h, mp, mode=O.MODE_TUNED, threads=8))
    # multi-pass march: only the first pass starts at the eye
    gpu.set_compaction(True)
    try:
        _compare(gpu.render(cam, w, h, mp), O.render(buf, cam, w, h, mp, mode=O.MODE_TUNED, threads=8))
    finally:
        gpu.set_compaction(False)
    # a pixel whose direction is the zero vector (image plane origin = eye): that ray's direction is NaN, its
    # lane does not stand at the eye, and its wavefront must not take the shared step for it
    import copy
    bad = copy.copy(cam)
    for a in range(3):
        bad.origin[a] = bad.eye[a]
    img, ref = gpu.render(bad, w, h, mp), O.render(buf, bad, w, h, mp, mode=O.MODE_TUNED, threads=8)
    assert np.array_equal(img["hit"], ref["hit"]) and np.array_equal(img["prim"], ref["prim"])
    ok = np.isfinite(ref["depth"]) & np.isfinite(img["depth"])
    assert np.array_equal(bits(img["depth"])[ok], bits(ref["depth"])[ok])
    assert np.array_equal(np.isnan(img["depth"]), np.isnan(ref["depth"]))

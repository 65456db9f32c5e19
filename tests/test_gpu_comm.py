"""The communicator entry points of the C ABI (sdfgpu_comm_*, sdfgpu_render_frame_comm) at world 1 on
the GPU: RCCL is loaded, a communicator is created, a frame goes through the partition -> packed
buffer -> assembly path and equals sdfgpu_render bit for bit.  (World > 1 needs one GPU per rank; the
tile deal and the assembly arithmetic it shares with this path are covered by
tests/test_partition_gloo.py and by the N-rank emulation in tests/test_gpu_render.py.)"""
import numpy as np
import pytest

import sdf_amd as S
from conftest import bits
from test_gpu_render import scene_camera

pytestmark = pytest.mark.gpu


def test_frame_through_a_world_1_communicator(gpu):
    import torch

    n, w, h = 300, 173, 99           # ragged: edge tiles on both axes
    buf = S.random_scene_buffer(n, 0xC0110)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf)
    want = gpu.render(cam, w, h, mp)
    comm = S.SdfComm(gpu, S.SdfComm.make_id(gpu), 0, 1)
    try:
        dev = torch.device("cuda:0")
        streams = [torch.cuda.current_stream(), torch.cuda.Stream(dev)]
        outs = []
        for st in streams:           # two frames in flight on two streams of one communicator
            o = (torch.zeros(w * h, dtype=torch.uint8, device=dev), torch.zeros(w * h, dtype=torch.float32, device=dev),
                 torch.zeros(w * h * 3, dtype=torch.float32, device=dev), torch.zeros(w * h, dtype=torch.int32, device=dev))
            comm.render_frame(cam, w, h, mp, 0, *(t.data_ptr() for t in o), st.cuda_stream)
            outs.append(o)
        torch.cuda.synchronize()
        for o in outs:
            assert np.array_equal(o[0].cpu().numpy(), want["hit"])
            assert np.array_equal(bits(o[1].cpu().numpy()), bits(want["depth"]))
            assert np.array_equal(bits(o[2].cpu().numpy().reshape(-1, 3)), bits(want["normal"]))
            assert np.array_equal(o[3].cpu().numpy().view(np.uint32), want["prim"])
        with pytest.raises(S.SdfGpuError):
            comm.render_frame(cam, w, h, mp, 3, 0, 0, 0, 0, 0)      # root outside the communicator
    finally:
        comm.close()


def _frame_tensors(torch, dev, w, h):
    return (torch.zeros(w * h, dtype=torch.uint8, device=dev), torch.zeros(w * h, dtype=torch.float32, device=dev),
            torch.zeros(w * h * 3, dtype=torch.float32, device=dev), torch.zeros(w * h, dtype=torch.int32, device=dev))


def _same_frame(o, want):
    assert np.array_equal(o[0].cpu().numpy(), want["hit"])
    assert np.array_equal(bits(o[1].cpu().numpy()), bits(want["depth"]))
    assert np.array_equal(bits(o[2].cpu().numpy().reshape(-1, 3)), bits(want["normal"]))
    assert np.array_equal(o[3].cpu().numpy().view(np.uint32), want["prim"])


def test_exchange_branch_runs_through_a_send_to_self(gpu, monkeypatch):
    """The world > 1 branch of sdfgpu_render_frame_comm -- ncclGroupStart, ncclRecv into the gather buffer,
    ncclSend of the packed share, ncclGroupEnd, untile from the gathered copy -- executed on the one GPU there is:
    with SDFGPU_COMM_SELF_EXCHANGE=1 the assembling rank exchanges its own share with itself.  Two frames on two
    streams, bit-identical to sdfgpu_render."""
    import torch

    n, w, h = 300, 173, 99
    buf = S.random_scene_buffer(n, 0xC0110)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf)
    want = gpu.render(cam, w, h, mp)
    monkeypatch.setenv("SDFGPU_COMM_SELF_EXCHANGE", "1")
    comm = S.SdfComm(gpu, S.SdfComm.make_id(gpu), 0, 1)
    monkeypatch.delenv("SDFGPU_COMM_SELF_EXCHANGE")
    try:
        dev = torch.device("cuda:0")
        streams = [torch.cuda.current_stream(), torch.cuda.Stream(dev)]
        outs = []
        for rep in range(2):             # the second round reuses each stream's staging
            for st in streams:
                o = _frame_tensors(torch, dev, w, h)
                comm.render_frame(cam, w, h, mp, 0, *(t.data_ptr() for t in o), st.cuda_stream)
                outs.append(o)
        torch.cuda.synchronize()
        for o in outs:
            _same_frame(o, want)
        # another frame size on the same stream: staging is re-made, the exchange still lines up
        w2, h2 = 64, 40
        cam2, mp2 = scene_camera(n, w2, h2)
        want2 = gpu.render(cam2, w2, h2, mp2)
        o = _frame_tensors(torch, dev, w2, h2)
        comm.render_frame(cam2, w2, h2, mp2, 0, *(t.data_ptr() for t in o), streams[0].cuda_stream)
        torch.cuda.synchronize()
        _same_frame(o, want2)
    finally:
        comm.close()


def test_local_failure_still_takes_part_in_the_exchange(monkeypatch):
    """A rank whose own march fails (here: no tree uploaded) must not leave its peers waiting: the exchange is
    still made -- its share travels as misses -- and the error comes back afterwards."""
    import torch

    g = S.SdfGpu(0)
    monkeypatch.setenv("SDFGPU_COMM_SELF_EXCHANGE", "1")
    comm = S.SdfComm(g, S.SdfComm.make_id(g), 0, 1)
    monkeypatch.delenv("SDFGPU_COMM_SELF_EXCHANGE")
    try:
        w, h = 40, 24
        cam, mp = scene_camera(10, w, h)
        dev = torch.device("cuda:0")
        o = _frame_tensors(torch, dev, w, h)
        o[0].fill_(7)
        with pytest.raises(S.SdfGpuError) as ei:
            comm.render_frame(cam, w, h, mp, 0, *(t.data_ptr() for t in o), 0)
        assert ei.value.status == S._ffi.ERR_NO_TREE
        torch.cuda.synchronize()                     # nothing hangs
        assert int(o[0].max().item()) == 0           # the assembled frame is all misses
        # the communicator is still usable once the rank has a tree
        g.upload_tree(S.random_scene_buffer(10, 3))
        want = g.render(cam, w, h, mp)
        comm.render_frame(cam, w, h, mp, 0, *(t.data_ptr() for t in o), 0)
        torch.cuda.synchronize()
        _same_frame(o, want)
    finally:
        comm.close()
        g.close()


def test_in_process_clique_of_one(gpu):
    """sdfgpu_comm_create_all (ncclCommInitAll) + sdfgpu_render_frame_all with the one device there is."""
    import torch

    n, w, h = 200, 96, 56
    buf = S.random_scene_buffer(n, 0xA11)
    cam, mp = scene_camera(n, w, h)
    gpu.upload_tree(buf)
    want = gpu.render(cam, w, h, mp)
    comms = S.SdfComm.create_all([gpu])
    try:
        dev = torch.device("cuda:0")
        o = _frame_tensors(torch, dev, w, h)
        st = torch.cuda.Stream(dev)
        S.SdfComm.render_frame_all(comms, cam, w, h, mp, 0, *(t.data_ptr() for t in o), streams=[st.cuda_stream])
        torch.cuda.synchronize()
        _same_frame(o, want)
        g2 = S.SdfGpu(0)
        try:
            with pytest.raises(S.SdfGpuError):       # two ranks of a clique on one device
                S.SdfComm.create_all([gpu, g2])
        finally:
            g2.close()
    finally:
        for m in comms:
            m.close()


def test_context_closes_its_communicators_first():
    g = S.SdfGpu(0)
    comm = S.SdfComm(g, S.SdfComm.make_id(g), 0, 1)
    g.close()                 # closes comm first (sdfgpu.h: communicators are destroyed before their context)
    assert comm._h is None
    comm.close()              # and closing again is harmless


def test_sharded_queries_equal_the_unsharded_call(gpu):
    """sdfgpu_nearest_sharded / sdfgpu_knn_sharded over three contexts (all on the one GPU): the batch is cut into
    contiguous slices, every slice lands at its own place, results are those of the single-context call."""
    n = 700
    buf = S.random_scene_buffer(n, 0x5A4D)
    L = S.default_extent(n)
    rng = np.random.default_rng(11)
    pts = np.ones((10_007, 4), np.float32)
    pts[:, :3] = rng.uniform(-1.2 * L, 1.2 * L, (len(pts), 3)).astype(np.float32)
    gpu.upload_tree(buf)
    others = [S.SdfGpu(0), S.SdfGpu(0)]
    try:
        for g in others:
            g.upload_tree(buf)
        d0, i0 = gpu.nearest(pts)
        d1, i1 = S.nearest_sharded([gpu] + others, pts)
        assert np.array_equal(bits(d0), bits(d1)) and np.array_equal(i0, i1)
        k0, j0 = gpu.knn(pts, 8)
        k1, j1 = S.knn_sharded([gpu] + others, pts, 8)
        assert np.array_equal(bits(k0), bits(k1)) and np.array_equal(j0, j1)
        # fewer points than contexts, and an empty batch
        d2, i2 = S.nearest_sharded([gpu] + others, pts[:2])
        assert np.array_equal(bits(d2), bits(d0[:2])) and np.array_equal(i2, i0[:2])
        d3, _ = S.knn_sharded([gpu] + others, pts[:0], 4)
        assert d3.shape == (0, 4)
        bad = pts[:5].copy()
        bad[3, 3] = 2.0
        with pytest.raises(S.SdfGpuError) as ei:
            S.nearest_sharded([gpu] + others, bad)
        assert ei.value.status == S._ffi.ERR_POINT_W
        fresh = S.SdfGpu(0)
        try:
            with pytest.raises(S.SdfGpuError) as ei:
                S.nearest_sharded([gpu, fresh], pts[:10])
            assert ei.value.status == S._ffi.ERR_NO_TREE
        finally:
            fresh.close()
    finally:
        for g in others:
            g.close()

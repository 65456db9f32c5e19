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

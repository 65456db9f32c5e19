"""Multi-rank path on CPU: two gloo ranks run the framebuffer partitioner
(tile deal, packed planes, all-gather, assembly) and the query slicer."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdf_amd import partition as P  # noqa: E402


def tile_pixels(width, height, tile):
    """(row-major pixel index or -1) of the 64 pixels of one tile -- what
    render_kernel / untile_kernel compute on the device."""
    tx, _ = P.tile_grid(width, height)
    x0, y0 = (tile % tx) * 8, (tile // tx) * 8
    out = np.full(64, -1, np.int64)
    for lane in range(64):
        x, y = x0 + (lane & 7), y0 + (lane >> 3)
        if x < width and y < height:
            out[lane] = y * width + x
    return out


def frame(width, height):
    rng = np.random.default_rng(5)
    n = width * height
    return {
        "hit": rng.integers(0, 2, n).astype(np.uint8),
        "depth": rng.random(n).astype(np.float32),
        "normal": rng.random((n, 3)).astype(np.float32),
        "prim": rng.integers(0, 1000, n).astype(np.int32),
    }


def worker(rank, world, port, width, height, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f = frame(width, height)
        tiles = P.n_tiles(width, height)
        first, stride, cnt = P.rank_tiles(rank, world, tiles)
        nbytes = P.packed_bytes(world, tiles)
        mine = torch.zeros(nbytes, dtype=torch.uint8)
        hit, depth, normal, prim = P.planes(mine, world, tiles)
        for k in range(cnt):                         # what sdfgpu_render_tiles_dev writes
            pix = tile_pixels(width, height, first + k * stride)
            ok = pix >= 0
            sl = slice(k * 64, (k + 1) * 64)
            hit[sl][ok] = torch.from_numpy(f["hit"][pix[ok]])
            depth[sl][ok] = torch.from_numpy(f["depth"][pix[ok]])
            normal.view(-1, 3)[sl][ok] = torch.from_numpy(f["normal"][pix[ok]])
            prim[sl][ok] = torch.from_numpy(f["prim"][pix[ok]])
        gathered = torch.empty(world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(gathered, mine)
        if rank == 0:                                # what sdfgpu_untile_dev does
            out = {k: np.zeros_like(v) for k, v in f.items()}
            seen = np.zeros(width * height, np.int32)
            for r in range(world):
                g_hit, g_depth, g_normal, g_prim = P.planes(gathered[r * nbytes:(r + 1) * nbytes], world, tiles)
                rf, rs, rc = P.rank_tiles(r, world, tiles)
                for k in range(rc):
                    pix = tile_pixels(width, height, rf + k * rs)
                    ok = pix >= 0
                    sl = slice(k * 64, (k + 1) * 64)
                    out["hit"][pix[ok]] = g_hit[sl].numpy()[ok]
                    out["depth"][pix[ok]] = g_depth[sl].numpy()[ok]
                    out["normal"][pix[ok]] = g_normal.view(-1, 3)[sl].numpy()[ok]
                    out["prim"][pix[ok]] = g_prim[sl].numpy()[ok]
                    seen[pix[ok]] += 1
            good = bool(np.all(seen == 1)) and all(np.array_equal(out[k], f[k]) for k in f)
            ret.put(good)
        # query slices tile [0, n) exactly
        n = 1003
        lo, hi = P.query_slice(rank, world, n)
        t = torch.tensor([hi - lo, lo], dtype=torch.int64)
        allv = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allv, t)
        if rank == 0:
            ret.put(sum(int(v[0]) for v in allv) == n and all(int(allv[r + 1][1]) == int(allv[r][1]) + int(allv[r][0]) for r in range(world - 1)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("width,height", [(64, 40), (61, 37)])
def test_two_rank_framebuffer_assembly(width, height):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, width, height, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True and ret.get(timeout=5) is True


def test_partition_arithmetic():
    for w, h in ((1920, 1080), (3840, 2160), (61, 37)):
        tiles = P.n_tiles(w, h)
        for world in (1, 2, 4, 8):
            counts = [P.rank_tiles(r, world, tiles)[2] for r in range(world)]
            assert sum(counts) == tiles and max(counts) - min(counts) <= 1
            assert P.packed_bytes(world, tiles) == max(counts) * 64 * 21
            assert P.packed_bytes(world, tiles) % 4 == 0

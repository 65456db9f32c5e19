"""GPU parity: sdfgpu_knn (K2) against the oracle's exhaustive top-k, and the
Morton-reordered query path against the unordered one."""
import numpy as np
import pytest

import sdf_amd as S
from oracle import oracle as O
from conftest import bits, make_points

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bounds", [S.BOUNDS_AABB, S.BOUNDS_OBB, S.BOUNDS_NONE, S.BOUNDS_BVH])
@pytest.mark.parametrize("n,k", [(1, 8), (3, 8), (5, 3), (64, 1), (200, 8), (200, 16), (1000, 8)])
def test_knn_matches_oracle(gpu, n, k, bounds):
    buf = S.random_scene_buffer(n, 0x4B00 + n)
    pts = make_points(3000, S.default_extent(n), seed=n + k)
    gpu.upload_tree(buf, bounds=bounds)
    d, i = gpu.knn(pts, k)
    d0, i0 = O.knn(buf, pts, k, threads=8)
    assert np.array_equal(bits(d), bits(d0))
    assert np.array_equal(i, i0)


def test_knn_first_column_is_nearest(gpu):
    buf = S.random_scene_buffer(300, 12)
    pts = make_points(5000, S.default_extent(300), 13)
    gpu.upload_tree(buf)
    d, i = gpu.knn(pts, 8)
    d1, i1 = gpu.nearest(pts)
    assert np.array_equal(bits(d[:, 0]), bits(d1)) and np.array_equal(i[:, 0], i1)
    assert np.all(np.diff(d, axis=1) >= 0)


def test_knn_duplicates_order_and_padding(gpu):
    t = S.Transform.from_translation((1.0, 2.0, 3.0))
    b = (S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t).operation(S.SdfUnion(0.0))
         .with_(S.SdfBuilder.primitive(S.SdfSphere(1.5)).transform(t))
         .with_(S.SdfBuilder.primitive(S.SdfBox((1, 2, 3))).transform(S.Transform.from_translation((9.0, 0, 0)))))
    buf = b.finalize().expanded().make_buffer()
    pts = make_points(512, 10.0, 3)
    gpu.upload_tree(buf)
    d, i = gpu.knn(pts, 8)
    d0, i0 = O.knn(buf, pts, 8)
    assert np.array_equal(bits(d), bits(d0)) and np.array_equal(i, i0)
    assert np.all(np.isposinf(d[:, 3:])) and np.all(i[:, 3:] == S.NO_PRIM)


def test_k_out_of_range(gpu):
    gpu.upload_tree(S.random_scene_buffer(4, 1))
    for k in (0, 17):
        with pytest.raises(S.SdfGpuError) as e:
            gpu.knn(make_points(4, 1.0, 1), k)
        assert e.value.status == S._ffi.ERR_TOO_LARGE


def test_reordering_is_invisible(gpu):
    buf = S.random_scene_buffer(500, 44)
    pts = make_points(50000, S.default_extent(500), 45, scale=3.0)   # some points far outside the scene box
    gpu.upload_tree(buf)
    gpu.set_reorder(True)
    d_on, i_on = gpu.nearest(pts)
    k_on = gpu.knn(pts, 4)
    gpu.set_reorder(False)
    d_off, i_off = gpu.nearest(pts)
    k_off = gpu.knn(pts, 4)
    gpu.set_reorder(True)
    assert np.array_equal(bits(d_on), bits(d_off)) and np.array_equal(i_on, i_off)
    assert np.array_equal(bits(k_on[0]), bits(k_off[0])) and np.array_equal(k_on[1], k_off[1])
    d0, i0 = O.nearest(buf, pts, O.MODE_TUNED, threads=8)
    assert np.array_equal(bits(d_on), bits(d0)) and np.array_equal(i_on, i0)


def test_chunked_pipeline_is_invisible(gpu):
    # batches of >= 2^20 points are cut into chunks whose Morton sort overlaps the previous chunk's walk
    # (SDFGPU_OPT_QUERY_PIPELINE); results must not depend on the chunking, including a ragged last chunk
    import torch
    from sdf_amd import _ffi

    n_prims, n = 2000, (1 << 20) + 12345
    buf = S.random_scene_buffer(n_prims, 0x91E)
    gpu.upload_tree(buf)
    pts = torch.from_numpy(make_points(n, S.default_extent(n_prims), seed=77)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    try:
        for chunks in (0, 4, 3):
            gpu.set_option(_ffi.OPT_QUERY_PIPELINE, chunks)
            d1 = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
            i1 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
            d8 = torch.full((n * 8,), -1.0, dtype=torch.float32, device="cuda")
            i8 = torch.full((n * 8,), -1, dtype=torch.int32, device="cuda")
            for _ in range(2):      # twice: the second call reuses scratch sets and events of the first
                gpu.nearest_dev(pts.data_ptr(), n, d1.data_ptr(), i1.data_ptr(), stream)
                gpu.knn_dev(pts.data_ptr(), n, 8, d8.data_ptr(), i8.data_ptr(), stream)
            torch.cuda.synchronize()
            res[chunks] = (d1, i1, d8, i8)
    finally:
        gpu.set_option(_ffi.OPT_QUERY_PIPELINE, 0)
    for chunks in (4, 3):
        for a, b in zip(res[0], res[chunks]):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f"chunks={chunks}"
    # and against the oracle on a sample
    sel = np.arange(0, n, 4099)
    d0, i0 = O.nearest(buf, pts[sel].cpu().numpy(), O.MODE_TUNED, threads=8)
    assert np.array_equal(bits(res[4][0][sel].cpu().numpy()), bits(d0))
    assert np.array_equal(res[4][1][sel].cpu().numpy().view(np.uint32), i0)


def test_batches_in_flight_on_several_streams(gpu):
    # query batches of ONE context in flight on different streams: each stream has its own sort buffers
    # (keys, permutation, radix storage), so the batches cannot disturb each other
    import torch

    n_prims, n = 3000, 300_000
    buf = S.random_scene_buffer(n_prims, 0x57E)
    gpu.upload_tree(buf)
    L = S.default_extent(n_prims)
    streams = [torch.cuda.Stream() for _ in range(3)]
    sets = [torch.from_numpy(make_points(n - 1000 * j, L, seed=300 + j)).cuda() for j in range(3)]
    torch.cuda.synchronize()
    want = []
    for p in sets:                      # one after the other on the context's own stream
        m = p.shape[0]
        want.append((gpu.nearest(p.cpu().numpy()), gpu.knn(p.cpu().numpy(), 8)))
    outs = []
    for rnd in range(2):                # interleaved, nothing synchronised in between
        for j, (st, p) in enumerate(zip(streams, sets)):
            m = p.shape[0]
            d1 = torch.empty(m, dtype=torch.float32, device="cuda")
            i1 = torch.empty(m, dtype=torch.int32, device="cuda")
            d8 = torch.empty(m * 8, dtype=torch.float32, device="cuda")
            i8 = torch.empty(m * 8, dtype=torch.int32, device="cuda")
            gpu.nearest_dev(p.data_ptr(), m, d1.data_ptr(), i1.data_ptr(), st.cuda_stream)
            gpu.knn_dev(p.data_ptr(), m, 8, d8.data_ptr(), i8.data_ptr(), st.cuda_stream)
            outs.append((j, d1, i1, d8, i8))
    torch.cuda.synchronize()
    for j, d1, i1, d8, i8 in outs:
        (wd, wi), (wkd, wki) = want[j]
        assert np.array_equal(bits(d1.cpu().numpy()), bits(wd)) and np.array_equal(i1.cpu().numpy().view(np.uint32), wi)
        assert np.array_equal(bits(d8.cpu().numpy().reshape(-1, 8)), bits(wkd))
        assert np.array_equal(i8.cpu().numpy().view(np.uint32).reshape(-1, 8), wki)


@pytest.mark.parametrize("shape", ["uniform", "one_cell", "clustered", "outliers", "ragged_small"])
def test_batch_sort_yields_a_permutation_on_any_distribution(gpu, shape):
    """The library's own batch sort (sort.hip: two 12-bit digits, buckets cut into chunks when a cell is crowded) must
    hand the walk a PERMUTATION whatever the distribution: outputs start as a sentinel, so an index that the
    permutation misses (or repeats, which makes it miss another) shows as an unwritten result."""
    import torch

    n_prims = 800
    buf = S.random_scene_buffer(n_prims, 0x50B7)
    L = S.default_extent(n_prims)
    gpu.upload_tree(buf)
    rng = np.random.default_rng(len(shape))
    n = {"uniform": 300_001, "one_cell": 150_000, "clustered": 400_000, "outliers": 200_000, "ragged_small": 2049}[shape]
    p = np.ones((n, 4), np.float32)
    if shape in ("uniform", "ragged_small"):
        p[:, :3] = rng.uniform(-1.2 * L, 1.2 * L, (n, 3))
    elif shape == "one_cell":            # every point in one cell of the curve: one bucket, many chunks
        p[:, :3] = np.float32(0.37 * L) + rng.uniform(0, 1e-3, (n, 3))
    elif shape == "clustered":           # 90 % of the batch in a ball of 1 % of the scene
        p[:, :3] = rng.uniform(-1.2 * L, 1.2 * L, (n, 3))
        m = int(0.9 * n)
        p[:m, :3] = rng.normal(0.2 * L, 0.01 * L, (m, 3))
    else:                                # half of the batch far outside the key box: clamped to its border cells
        p[:, :3] = rng.uniform(-1.2 * L, 1.2 * L, (n, 3))
        p[::2, :3] *= 40.0
    pts = torch.from_numpy(p).cuda()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for reorder in (True, False):
        gpu.set_reorder(reorder)
        d = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
        i = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        gpu.nearest_dev(pts.data_ptr(), n, d.data_ptr(), i.data_ptr(), st)
        kd = torch.full((n * 4,), -7.0, dtype=torch.float32, device="cuda")
        ki = torch.full((n * 4,), -7, dtype=torch.int32, device="cuda")
        gpu.knn_dev(pts.data_ptr(), n, 4, kd.data_ptr(), ki.data_ptr(), st)
        torch.cuda.synchronize()
        out[reorder] = tuple(t.cpu().numpy() for t in (d, i, kd, ki))
    gpu.set_reorder(True)
    for a, b in zip(out[True], out[False]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert (out[True][1] >= 0).all() and (out[True][3] >= 0).all()     # every slot written: no index missed
    d0, i0 = O.nearest(buf, p[:3000], O.MODE_TUNED, threads=8)
    assert np.array_equal(bits(out[True][0][:3000]), bits(d0)) and np.array_equal(out[True][1][:3000].view(np.uint32), i0)


def test_chunked_pipeline_with_two_caller_streams(gpu):
    """Two large batches in flight on two caller streams with SDFGPU_OPT_QUERY_PIPELINE on share the context's two
    pipeline sets and its helper stream: the helper must not re-sort a set while the OTHER stream's walk still reads
    its permutation (it waits for both sets' last walks before the first sort of a call)."""
    import torch
    from sdf_amd import _ffi

    n_prims, n = 1500, (1 << 20) + 777
    buf = S.random_scene_buffer(n_prims, 0x2577)
    gpu.upload_tree(buf)
    L = S.default_extent(n_prims)
    sets = [torch.from_numpy(make_points(n, L, seed=900 + j)).cuda() for j in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    want = []
    for p in sets:
        d = torch.empty(n, dtype=torch.float32, device="cuda")
        i = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.nearest_dev(p.data_ptr(), n, d.data_ptr(), i.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        want.append((d, i))
    try:
        gpu.set_option(_ffi.OPT_QUERY_PIPELINE, 3)
        outs = []
        torch.cuda.synchronize()
        for rep in range(3):
            for j, (p, st) in enumerate(zip(sets, streams)):
                d = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
                i = torch.full((n,), -1, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize() if rep == 0 and j == 0 else None
                gpu.nearest_dev(p.data_ptr(), n, d.data_ptr(), i.data_ptr(), st.cuda_stream)
                outs.append((j, d, i))
        torch.cuda.synchronize()
    finally:
        gpu.set_option(_ffi.OPT_QUERY_PIPELINE, 0)
    for j, d, i in outs:
        assert torch.equal(d.view(torch.int32), want[j][0].view(torch.int32)) and torch.equal(i, want[j][1])

"""Space-filling-curve keys of sdf_amd.workloads (the caller-side presort of bench.py's kNN leg; the
same arithmetic as sort.hip's key kernel): the Hilbert index visits every cell once and consecutive
cells are face neighbours; the Morton code does not have that property (which is why the library
orders query batches along the Hilbert curve)."""
import numpy as np

from sdf_amd import workloads as W


def lattice(n):
    ii = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
    return ii, ((ii + 0.5) / n * 2 - 1).astype(np.float64)


def test_hilbert_index_is_a_continuous_bijection():
    n = 32
    ii, xyz = lattice(n)
    k = W.hilbert_keys(xyz, 1.0)
    assert len(np.unique(k >> 15)) == n ** 3          # 5 of the 10 bits per axis are in use
    order = np.argsort(k, kind="stable")
    step = np.abs(np.diff(ii[order], axis=0)).sum(1)
    assert step.max() == 1 and step.min() == 1


def test_morton_code_jumps():
    ii, xyz = lattice(16)
    order = np.argsort(W.morton_keys(xyz, 1.0), kind="stable")
    assert np.abs(np.diff(ii[order], axis=0)).sum(1).max() > 1


def test_presorted_points_are_a_permutation_of_the_unsorted_ones():
    wl = W.CONFIGS[1]
    a = W.query_points(wl, 5000)
    for curve in ("hilbert", "morton"):
        b = W.query_points(wl, 5000, morton=True, curve=curve)
        assert np.array_equal(np.sort(a.view(np.uint32), axis=0), np.sort(b.view(np.uint32), axis=0))

"""The five BASELINE.json configurations as data (SURVEY.md 8d recipes).

Scene(N, seed): N random sphere/box primitives under one flat union, extent
L = 50 (N/1000)^(1/3).  Camera: pinhole at (0, 0, -2.5 L) looking at the
origin, up +Y, vertical fov pi/4 (geometry of src/main__.rs:261-264).  March:
128 steps, eps 1e-3, t_max 6 L, normal h 1e-4 (src/main__.rs:152).
"""
import math
from dataclasses import dataclass

import numpy as np

from .gpu import camera_look_at, march_params
from .scene import default_extent, random_scene_buffer


@dataclass(frozen=True)
class Workload:
    name: str
    n_prims: int
    seed: int
    width: int = 0
    height: int = 0
    n_points: int = 0
    k: int = 1


CONFIGS = {
    1: Workload("config1: 1k prims, 64Ki nearest queries", 1000, 0x5DF00001, n_points=65536),
    2: Workload("config2: 1080p frame, 1k prims", 1000, 0x5DF00002, 1920, 1080),
    3: Workload("config3: 1080p frame, 10k prims", 10000, 0x5DF00003, 1920, 1080),
    4: Workload("config4: 4K frame, 100k prims", 100000, 0x5DF00004, 3840, 2160),
    5: Workload("config5: kNN k=8, 10M points vs 100k prims", 100000, 0x5DF00005, n_points=10_000_000, k=8),
}


def scene(wl: Workload):
    return random_scene_buffer(wl.n_prims, wl.seed)


def camera(wl: Workload, width=None, height=None):
    w, h = width or wl.width, height or wl.height
    L = default_extent(wl.n_prims)
    return camera_look_at((0.0, 0.0, -2.5 * L), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), math.pi / 4, w, h)


def march(wl: Workload):
    L = default_extent(wl.n_prims)
    return march_params(max_steps=128, eps_hit=1e-3, t_min=0.0, t_max=6.0 * L, normal_h=1e-4)


def query_points(wl: Workload, n=None, morton=False, curve="hilbert"):
    """n points ~ U[-1.2 L, 1.2 L)^3, w = 1, from a seeded numpy generator.  morton=True: sorted along a
    space-filling curve by the caller (`curve`: "hilbert", the order the library itself uses, or "morton")."""
    n = n or wl.n_points
    L = default_extent(wl.n_prims)
    rng = np.random.default_rng(wl.seed)
    p = np.ones((n, 4), np.float32)
    p[:, :3] = rng.uniform(-1.2 * L, 1.2 * L, (n, 3)).astype(np.float32)
    if morton:
        keys = hilbert_keys(p[:, :3], L * 1.2) if curve == "hilbert" else morton_keys(p[:, :3], L * 1.2)
        p = p[np.argsort(keys, kind="stable")]
    return p


def _cells(xyz, half_extent):
    return np.clip(((xyz / half_extent) * 0.5 + 0.5) * 1024.0, 0, 1023).astype(np.uint64)


def _spread10(v):
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_keys(xyz, half_extent):
    """30-bit Morton code of points in [-half_extent, half_extent)^3."""
    q = _cells(xyz, half_extent)
    return (_spread10(q[:, 0]) | (_spread10(q[:, 1]) << 1) | (_spread10(q[:, 2]) << 2)).astype(np.uint32)


def hilbert_keys(xyz, half_extent):
    """30-bit Hilbert index of the same cells (Skilling 2004, axes -> transposed index; the arithmetic
    of sort.hip's key kernel, vectorised)."""
    q = _cells(xyz, half_extent)
    X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    Q = 1 << 9
    while Q > 1:
        P = Q - 1
        for a in range(3):
            hit = (X[a] & Q) != 0
            t = (X[0] ^ X[a]) & P
            X0 = np.where(hit, X[0] ^ P, X[0] ^ t)
            if a:
                X[a] = np.where(hit, X[a], X[a] ^ t)
            X[0] = X0
        Q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = np.zeros_like(X[0])
    Q = 1 << 9
    while Q > 1:
        t = np.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for a in range(3):
        X[a] ^= t
    return ((_spread10(X[0]) << 2) | (_spread10(X[1]) << 1) | _spread10(X[2])).astype(np.uint32)


# Algorithmic fp32 operations per unit of work of the AABB-mode kernels
# (add/sub/mul/max/min/abs/compare/sqrt = 1 each; derivation in DESIGN.md).
FLOP_INNER = 40      # two AABB distance^2 + threshold + two compares
FLOP_SPHERE = 27     # 3x4 mat-vec (18) + length3 (6) + sub + better-than test (2)
FLOP_BOX = 39        # 3x4 mat-vec (18) + box formula (19) + better-than test (2)
FLOP_QUERY = 16      # ray point (6) + margin (7) + hit / t_max tests and t update (3)
FLOP_NORMAL_TAP = 6  # 3 point offsets + 3 accumulations per tap (leaf evals counted as sphere/box)
FLOP_CELL = 9        # a march step decided on the candidate grid: cell of the point = 3 x (sub, mul, floor)


def algorithmic_flops(stats: dict) -> int:
    """Algorithmic flops of one launch from the counting kernel's counters
    (lane-level: work done for lanes that were alive)."""
    per_step = max(stats["steps"], stats["queries"])   # render: every march step; nearest: every query
    return (stats["inner_visits"] * FLOP_INNER + stats["sphere_evals"] * FLOP_SPHERE
            + stats["box_evals"] * FLOP_BOX + per_step * FLOP_QUERY + stats.get("list_steps", 0) * FLOP_CELL)


def algorithmic_flops_needed(stats: dict) -> int:
    """The same count restricted to the units a lane itself needed: internal nodes its own bound asked
    for, seed evaluations, leaves its own bound asked for (the wavefront visits the union of what its
    lanes need; lanes that only ride along are not counted here)."""
    per_step = max(stats["steps"], stats["queries"])
    return (stats["inner_needed"] * FLOP_INNER + stats["sphere_needed"] * FLOP_SPHERE
            + stats["box_needed"] * FLOP_BOX + per_step * FLOP_QUERY + stats.get("list_steps", 0) * FLOP_CELL)

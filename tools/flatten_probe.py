#!/usr/bin/env python3
"""Host flatten timing: SdfNode::expanded (union split recursion with PCA boxes) and make_buffer, 100k primitives."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
t0 = time.perf_counter()
node = S.random_scene(n, 0x5DF00005)
t1 = time.perf_counter()
best = [1e9, 1e9]
for _ in range(5):
    a = time.perf_counter()
    ex = node.expanded()
    b = time.perf_counter()
    buf = ex.make_buffer()
    c = time.perf_counter()
    best = [min(best[0], (b - a) * 1e3), min(best[1], (c - b) * 1e3)]
print(f"threads {os.environ.get('SDFGPU_BUILD_THREADS', 'default')}: scene {1e3 * (t1 - t0):.1f} ms, expanded {best[0]:.1f} ms, make_buffer {best[1]:.1f} ms")

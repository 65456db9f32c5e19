#!/usr/bin/env python3
"""Upload timing: the host half alone (sdfgpu_compact_host: validation + SAH build) against the whole
sdfgpu_upload_tree (+ allocation and copy), best of 5, versus SDFGPU_BUILD_THREADS."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
buf = S.random_scene_buffer(n, 0x5DF00005)
gpu = S.SdfGpu(0)
gpu.upload_tree(buf)
best = [1e9, 1e9]
for _ in range(5):
    a = time.perf_counter()
    S.compact_host(buf)
    b = time.perf_counter()
    gpu.upload_tree(buf)
    c = time.perf_counter()
    best = [min(best[0], (b - a) * 1e3), min(best[1], (c - b) * 1e3)]
print(f"threads {os.environ.get('SDFGPU_BUILD_THREADS', 'default')}: compact_host {best[0]:.1f} ms, upload_tree {best[1]:.1f} ms")

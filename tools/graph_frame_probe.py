#!/usr/bin/env python3
"""A frame (sdfgpu_render_dev: tile schedule + eye walk + march) captured into a HIP graph after one warm-up call
on the same stream; replays timed against plain calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402


def run(log=print, config=3, frames=50):
    wl = W.CONFIGS[config]
    gpu = S.SdfGpu(0)
    gpu.upload_tree(W.scene(wl))
    w, h = wl.width, wl.height
    cam, mp = W.camera(wl, w, h), W.march(wl)
    outs = [torch.empty(w * h, dtype=torch.uint8, device="cuda"), torch.empty(w * h, dtype=torch.float32, device="cuda"),
            torch.empty(w * h * 3, dtype=torch.float32, device="cuda"), torch.empty(w * h, dtype=torch.int32, device="cuda")]
    ref = [torch.empty_like(o) for o in outs]
    s = torch.cuda.Stream()

    def frame(bufs, stream):
        gpu.render_dev(cam, w, h, mp, *(o.data_ptr() for o in bufs), stream)

    with torch.cuda.stream(s):
        for _ in range(3):
            frame(ref, s.cuda_stream)
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        frame(outs, torch.cuda.current_stream().cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(a.view(torch.uint8), b.view(torch.uint8)) for a, b in zip(outs, ref))
    log(f"replayed frame identical to a plain call: {same}")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / frames * 1e3
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(frames):
            frame(ref, s.cuda_stream)
    torch.cuda.synchronize()
    tp = (time.perf_counter() - t0) / frames * 1e3
    log(f"{frames} frames back to back on one stream: graph replay {tg:.3f} ms per frame, plain calls {tp:.3f} ms per frame")
    return same


if __name__ == "__main__":
    sys.exit(0 if run() else 1)

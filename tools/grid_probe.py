"""Design probe (not product, not a test): the bench frame with the candidate grid under different sizings --
work counters of the counting kernel and the isolated frame time.  Run on the GPU box:
  python tools/grid_probe.py [config] ; sizing through SDFGPU_GRID_* in the environment (read at first upload)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    wl = W.CONFIGS[cfg]
    w, h = wl.width, wl.height
    buf = W.scene(wl)
    cam, mp = W.camera(wl, w, h), W.march(wl)
    dev = torch.device("cuda:0")
    out = {}
    for grid in (0, 1):
        g = S.SdfGpu(0)
        g.set_candidate_grid(bool(grid))
        t0 = time.perf_counter()
        info = g.upload_tree(buf)
        up = (time.perf_counter() - t0) * 1e3
        o = (torch.empty(w * h, dtype=torch.uint8, device=dev), torch.empty(w * h, dtype=torch.float32, device=dev),
             torch.empty(w * h * 3, dtype=torch.float32, device=dev), torch.empty(w * h, dtype=torch.int32, device=dev))
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(5):
            g.render_dev(cam, w, h, mp, *(t.data_ptr() for t in o), st)
        torch.cuda.synchronize()
        ms = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.render_dev(cam, w, h, mp, *(t.data_ptr() for t in o), st)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        s = g.render_stats(cam, w, h, mp)
        out[f"grid{grid}"] = {
            "frame_ms_min": min(ms), "frame_ms_avg": sum(ms) / len(ms), "upload_ms": up, "grid_cells": int(info.grid_cells),
            "device_MB": info.device_bytes / 1e6, "walks_per_ray": s["queries"] / max(1, s["rays"]),
            "wave_steps": s["wave_steps"], "wave_walks": s["wave_walks"], "wave_lists": s["wave_lists"],
            "wave_list_evals": s["wave_list_evals"], "evals_per_list_pass": s["wave_list_evals"] / max(1, s["wave_lists"]),
            "list_steps": s["list_steps"], "steps": s["steps"], "wave_inner": s["wave_inner"], "wave_leaf": s["wave_leaf"],
            "checksum": int(o[3].sum().item()) ^ int(o[0].sum().item()),
        }
        g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

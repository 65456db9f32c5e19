#!/usr/bin/env python3
"""One case of the config-5 query batch, alone, for a counter pass (tools/query_traffic.sh):
   tools/query_traffic.py <unsorted|presorted> <k: 1|8> [launches]
unsorted: the workload's points as generated, reordered inside the call (SDFGPU_OPT_REORDER on, the default);
presorted: the same points sorted along the Hilbert curve by the caller, reordering off."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sdf_amd as S  # noqa: E402
from sdf_amd import workloads as W  # noqa: E402


def main():
    case, k = sys.argv[1], int(sys.argv[2])
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    wl = W.CONFIGS[5]
    gpu = S.SdfGpu(0)
    gpu.upload_tree(W.scene(wl))
    n = 10_000_000
    dev = torch.device("cuda", 0)
    pts = torch.from_numpy(W.query_points(wl, n, morton=(case == "presorted"))).to(dev)
    gpu.set_reorder(case != "presorted")
    d_out = torch.empty(n * k, dtype=torch.float32, device=dev)
    i_out = torch.empty(n * k, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(launches):
        if k == 1:
            gpu.nearest_dev(pts.data_ptr(), n, d_out.data_ptr(), i_out.data_ptr(), stream)
        else:
            gpu.knn_dev(pts.data_ptr(), n, k, d_out.data_ptr(), i_out.data_ptr(), stream)
        torch.cuda.synchronize()
    print(case, k, "done", float(d_out[:8].sum()))


if __name__ == "__main__":
    main()

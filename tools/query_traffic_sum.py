#!/usr/bin/env python3
"""Sum tools/query_traffic.sh's passes: MB fetched / written per batch (all kernels of the call: curve keys,
radix sort, walk), the last launch of each run (warm), against the batch's algorithmic bytes."""
import collections
import csv
import glob
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qtraffic"
out = {}
for case in ("unsorted", "presorted"):
    for k in (1, 8):
        row = {"algorithmic_MB": (16 + 8 * k) * 10.0}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            name = f"{case}_k{k}_{ctr}"
            paths = glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)
            if not paths:
                continue
            per = collections.defaultdict(list)   # kernel -> values in dispatch order
            with open(paths[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] != ctr:
                        continue
                    kn = r["Kernel_Name"]
                    # round 3: the library's own two-digit sort (rs_keys_hist computes the keys as well) in place of morton_kernel + rocprim
                    kn = "walk" if ("nearest_kernel" in kn or "knn_kernel" in kn) else ("keys" if ("morton_kernel" in kn or "rs_keys_hist" in kn) else ("sort" if ("rocprim" in kn or "::rs_" in kn) else None))
                    if kn:
                        per[kn].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            # the last batch: the last walk launch and the key / sort launches after the walk launch before it
            walks = sorted(per["walk"])
            last_walk = walks[-1][0]
            prev_walk = walks[-2][0] if len(walks) > 1 else -1
            for kn, v in per.items():
                kb = sum(val for d, val in v if prev_walk < d <= last_walk)
                row[f"{ctr}_{kn}_MB"] = round(kb * 1024 / 1e6, 1)   # KiB -> MB
        row["fetched_MB"] = round(sum(v for q, v in row.items() if q.startswith("FETCH_SIZE_")), 1)
        row["written_MB"] = round(sum(v for q, v in row.items() if q.startswith("WRITE_SIZE_")), 1)
        row["ratio"] = round((row["fetched_MB"] + row["written_MB"]) / row["algorithmic_MB"], 2)
        out[f"{case}_k{k}"] = row
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into the files kept under profiles/.

usage: tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>
Writes kernel_stats.csv (rocprofv3 --kernel-trace --stats, names shortened), pmc_<kernel>.json
(mean counter value per launch, per kernel of this library) and bench_line.json.
"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    m = re.match(r"(?:void )?([\w:]+(?:<[^(]*?>)?)\(", name)
    s = m.group(1) if m else name
    return s if len(s) < 160 else s[:157] + "..."


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(src, "trace", "trace_kernel_stats.csv")) as f:
        rows = list(csv.DictReader(f))
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in ("pmc_sq1", "pmc_sq2", "pmc_fetch", "pmc_write"):
        path = os.path.join(src, p, p + "_counter_collection.csv")
        if not os.path.exists(path):
            continue
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if "sdfgpu::" not in k:
                    continue
                per_kernel[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, ctr in per_kernel.items():
        out[k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in sorted(ctr.items())}
        c = out[k]
        d = {}
        g = lambda n: c[n]["mean_per_launch"] if n in c else None
        if g("SQ_INSTS_VALU") and g("SQ_ACTIVE_INST_VALU"):
            d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES") if g("SQ_WAVES") else None
        if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
            d["note"] = "SQ_* are summed over shader engines / XCDs as rocprofv3 reports them"
        # HBM traffic: FETCH_SIZE is in 32-B units? the guide: FETCH_SIZE in kilobytes on gfx950 (rocprofv3 derived)
        out[k]["_derived"] = d
    with open(os.path.join(dst, "pmc_by_kernel.json"), "w") as f:
        json.dump(out, f, indent=1)
    tj = os.path.join(src, "trace.json")
    if os.path.exists(tj):
        lines = [l for l in open(tj) if l.startswith("{")]
        if lines:
            with open(os.path.join(dst, "bench_line.json"), "w") as f:
                f.write(lines[-1])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

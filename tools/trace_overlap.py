#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV of bench.py at its default pipeline depth into the evidence
that frames on different streams overlap: one row per render launch (queue, start, end relative to the
first, ms) and the share of each launch's duration during which another render launch was running.

usage: tools/trace_overlap.py <..._kernel_trace.csv> <out.csv>"""
import csv
import sys


def main(src, dst):
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            if "render_pass_kernel<1, false, true" in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r.get("Stream_Id", "")))
    rows.sort()
    t0 = rows[0][0]
    out = []
    for i, (s, e, q, st) in enumerate(rows):
        ov = 0
        for j, (s2, e2, _, _) in enumerate(rows):
            if j != i:
                ov += max(0, min(e, e2) - max(s, s2))
        out.append((i, q, st, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, min(1.0, ov / max(1, e - s))))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["launch", "queue_id", "stream_id", "start_ms", "end_ms", "duration_ms", "overlapped_share"])
        for r in out:
            w.writerow([r[0], r[1], r[2], f"{r[3]:.3f}", f"{r[4]:.3f}", f"{r[5]:.3f}", f"{r[6]:.3f}"])
    tail = out[len(out) // 2:]
    print(f"{len(out)} render launches on queues {sorted(set(r[1] for r in out))}; second half: "
          f"mean duration {sum(r[5] for r in tail) / len(tail):.3f} ms, mean overlapped share {sum(r[6] for r in tail) / len(tail):.2f}, "
          f"throughput {(tail[-1][4] - tail[0][3]) / len(tail):.3f} ms per frame")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

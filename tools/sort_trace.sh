#!/bin/bash
# per-kernel times of the unsorted 10 M point query (tools/sort_probe.py) under rocprofv3: tools/sort_trace.sh <tag>
set -o pipefail
TAG=${1:-t}; OUT=gpurun_out/sort_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 200 python3 tools/sort_probe.py || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 tools/sort_probe.py > $OUT/out.txt 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/trace/trace_kernel_stats.csv")))
for r in rows[:12]:
    print("%-60s calls %5s avg %9.1f us  min %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY

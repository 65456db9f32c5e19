#!/bin/bash
# Kernel trace of the kNN leg alone (design probe): tools/knn_trace.sh <tag>
set -o pipefail
TAG=${1:-t}; OUT=gpurun_out/knn_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-variants --knn-steps 10 > $OUT/line.json 2> $OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/trace/trace_kernel_stats.csv")))
for r in rows[:16]:
    print("%-70s calls %5s avg %9.1f us  min %9.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY

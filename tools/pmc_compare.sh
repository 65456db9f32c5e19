#!/bin/bash
# PMC pass over one bench variant: tools/pmc_compare.sh <tag> <bench args...>
set -o pipefail
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o a -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-knn "$@" > $OUT/a.json 2>$OUT/a.err || { tail -3 $OUT/a.err; exit 1; }
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$OUT/a/a_counter_collection.csv")))
agg=collections.defaultdict(list)
for r in rows:
    if 'render_pass_kernel<1, false' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
print("$TAG", {k: "%.3g"%(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY

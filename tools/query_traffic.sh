#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every kernel of one query batch, case by case (tools/query_traffic.py), one
# counter pass per run.  Output: gpurun_out/qtraffic/<case>_k<k>_<pass>/..., summary by tools/query_traffic_sum.py
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/qtraffic
mkdir -p $OUT
for case in unsorted presorted; do
  for k in 1 8; do
    for pass in FETCH_SIZE WRITE_SIZE; do
      name=${case}_k${k}_${pass}
      timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/$name -o $name -- python3 tools/query_traffic.py $case $k 4 > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; exit 1; }
      echo "$name ok"
    done
  done
done

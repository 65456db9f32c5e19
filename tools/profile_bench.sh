#!/bin/bash
# Profiling recipe for bench.py on the GPU box (one pass per run, see
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage: tools/profile_bench.sh <tag> [bench args]
# Writes gpurun_out/prof_<tag>/...; copy the summaries you keep into profiles/.
set -o pipefail
TAG=${1:-r1}; shift
ARGS=${@:---steps 10 --warmup 2 --no-cpu --no-variants}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, rocprof flags...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d $OUT/$name -o $name -- python3 bench.py $ARGS > $OUT/$name.json 2> $OUT/$name.err || { echo "pass $name failed"; tail -5 $OUT/$name.err; return 1; }
}
run trace --kernel-trace --stats &&
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY &&
run pmc_sq2 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE &&
run pmc_fetch --pmc FETCH_SIZE &&
run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum &&
# the same bench at its DEFAULT pipeline depth (two frames in flight on two streams): kernel trace only
timeout -s KILL 240 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-knn --no-variants > $OUT/p2.json 2> $OUT/p2.err || echo "pass p2 failed"
find $OUT -name "*.csv" | head -40

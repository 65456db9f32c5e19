#!/bin/bash
# Soak run of the randomised GPU parity check (tools/fuzz_gpu.py, heavy flavour): several seeds, each with
# the default walk and with the prefetching copy of the walk forced (SDFGPU_PREFETCH=1), which small trees
# never select by themselves.  One line per run goes to gpurun_out/soak.log; stops at the first failure.
#   tools/soak_gpu.sh <first seed> <number of seeds> [iterations per seed]
set -u
first=${1:-100}; count=${2:-8}; iters=${3:-40}
mkdir -p gpurun_out
for ((s = first; s < first + count; ++s)); do
  for pf in default 1; do
    if [ "$pf" = default ]; then unset SDFGPU_PREFETCH; else export SDFGPU_PREFETCH=$pf; fi
    t0=$(date +%s)
    timeout -k 10 300 python3 -u tools/fuzz_gpu.py "$s" "$iters" heavy > gpurun_out/soak_last.txt 2>&1
    rc=$?
    echo "seed $s prefetch $pf iters $iters rc $rc $(( $(date +%s) - t0 )) s: $(tail -n 1 gpurun_out/soak_last.txt)" | tee -a gpurun_out/soak.log
    if [ $rc -ne 0 ]; then tail -n 20 gpurun_out/soak_last.txt; exit $rc; fi
  done
done

#!/bin/bash
# FETCH_SIZE and WRITE_SIZE of every kernel of a c3 batch loop, one rocprofv3 pass per counter
# (run from the repository root on the GPU box); extra arguments go to bench.py.
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout -s KILL 150 rocprofv3 --pmc $c -d $R/gpurun_out/pmc_$c -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-recall "$@" > $R/gpurun_out/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -3 $R/gpurun_out/pmc_$c.log; exit 1; }
done
echo done

#!/bin/bash
# Calibrates the fabric-side read counters on the scan's own access shape: a known byte count
# (8M rows x 256 B, each row once per pass, table far larger than the Infinity Cache) pulled as
# MFMA operands (shape 0: one instruction = 16 rows x 64 B) and as whole rows (shape 1).
R=$PWD
cd /tmp && export TMPDIR=/tmp
for shape in 0 1; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum -d $R/gpurun_out/cal$shape -o p --output-format csv -- $R/tools/bin/rowload_bench 8388608 2 $shape > $R/gpurun_out/cal$shape.log 2>&1 || { echo "shape $shape failed"; tail -3 $R/gpurun_out/cal$shape.log; exit 1; }
done
echo done

#!/bin/bash
# usage (GPU box, repository root): bash tools/timeline_of.sh <tag> [bench.py arguments]
# (TL_LIB=<library> in the environment: that build of the library instead of the package's)
# kernel trace of a short bench run; prints the kernels of one late batch (begin, gap, duration) into gpurun_out/tl_<tag>.txt
T=$1; shift
R=$PWD
mkdir -p $R/gpurun_out
PROG=$R/bench.py
if [ -n "$TL_LIB" ]; then PROG="$R/tools/run_with_lib.py $R/$TL_LIB"; fi
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/kt_$T
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$T -o kt --output-format csv -- python3 $PROG --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 --no-gather-check --steps 8 "$@" > $R/gpurun_out/tl_$T.log 2>&1 || { tail -5 $R/gpurun_out/tl_$T.log; exit 1; }
cd $R
python3 tools/last_batch.py $(find gpurun_out/kt_$T -name 'kt_kernel_trace.csv' | head -1) 8 > gpurun_out/tl_$T.txt
rm -rf gpurun_out/kt_$T
cat gpurun_out/tl_$T.txt

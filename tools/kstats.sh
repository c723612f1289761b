#!/bin/bash
# usage (GPU box, repository root): bash tools/kstats.sh <tag> <library or -> [bench.py arguments]
# per-kernel average durations of a short bench run under rocprofv3 --kernel-trace, into gpurun_out/ks_<tag>.txt
T=$1; LIB=$2; shift; shift
R=$PWD
mkdir -p $R/gpurun_out
if [ "$LIB" = "-" ]; then LIB=similaritysearchbyrdf_amd/libdpf.so; fi
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/kt_$T
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/kt_$T -o kt --output-format csv -- python3 $R/tools/run_with_lib.py $R/$LIB --no-cpu-baseline --no-recall --no-host-to-host --no-out-of-sample --large-table-rows 0 --no-gather-check --steps 8 --warmup 4 "$@" > $R/gpurun_out/ks_$T.log 2>&1 || { tail -5 $R/gpurun_out/ks_$T.log; exit 1; }
cd $R
python3 tools/trace_summary.py $(find gpurun_out/kt_$T -name 'kt_kernel_trace.csv' | head -1) > gpurun_out/ks_$T.txt
rm -rf gpurun_out/kt_$T
head -24 gpurun_out/ks_$T.txt

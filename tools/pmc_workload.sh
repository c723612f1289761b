#!/bin/bash
# HBM-side traffic and matrix-core counters of every kernel of one workload's batch loop, one rocprofv3 pass per
# counter group, plus a kernel trace with statistics of the same command (run from the repository root on the
# GPU box): tools/pmc_workload.sh <tag> [bench.py arguments]   ->   gpurun_out/pmc_<tag>/{fetch,write,mfma,kt}
T=$1; shift
R=$PWD
O=$R/gpurun_out/pmc_$T
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 $*"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1 || { echo "pass fetch failed"; tail -3 $O/fetch.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/bench.py $ARGS > $O/write.log 2>&1 || { echo "pass write failed"; tail -3 $O/write.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o p --output-format csv -- python3 $R/bench.py $ARGS > $O/mfma.log 2>&1 || { echo "pass mfma failed"; tail -3 $O/mfma.log; exit 1; }
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py $ARGS > $O/kt.log 2>&1 || { echo "kernel trace failed"; tail -3 $O/kt.log; exit 1; }
echo "pmc $T done"

# usage: bash tools/pmc_scan_lib.sh <library> [bench.py arguments]: tools/pmc_scan.sh against another build of the library
R=$PWD
LIB=$1; shift
rm -rf $R/gpurun_out/pmc[0-9]*
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $R/gpurun_out/pmc$i -o p --output-format csv -- python3 $R/tools/run_with_lib.py $R/$LIB --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 "$@" > $R/gpurun_out/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc$i.log; exit 1; }
done
echo done

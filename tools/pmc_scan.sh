#!/bin/bash
# PMC passes over one workload's batch loop (run from the repository root on the GPU box); one pass per counter set:
# tools/pmc_scan.sh [bench.py arguments, e.g. --workload c4]   ->   gpurun_out/pmc1 .. pmc6, then tools/pmc_detail.py
R=$PWD
rm -rf $R/gpurun_out/pmc[0-9]*
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $R/gpurun_out/pmc$i -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 "$@" > $R/gpurun_out/pmc$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc$i.log; exit 1; }
done
echo done

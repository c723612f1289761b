#!/bin/bash
# instruction mix and wave-cycle breakdown of the kernels of one batch loop (run from the repository root on the GPU box):
# bash tools/pmc_insts.sh [bench.py arguments]; then python3 tools/pmc_insts_summary.py <kernel name part> ...
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD" \
           "SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_CYCLES"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_i$i -o p --output-format csv -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-recall --no-host-to-host --no-out-of-sample --large-table-rows 0 --no-gather-check "$@" > $R/gpurun_out/pmc_i$i.log 2>&1 || { tail -3 $R/gpurun_out/pmc_i$i.log; exit 1; }
done
echo done

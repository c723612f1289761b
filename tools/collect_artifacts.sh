#!/bin/bash
# Everything profiles/ is refreshed from, in one go on the GPU box (run from the repository root):
# GPU tests, the default bench, its rocprofv3 kernel trace + stats, the other configurations, the PMC passes.
R=$PWD
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/final_tests.log 2>&1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/final_c3.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt --output-format csv -- python3 $R/bench.py > $R/gpurun_out/final_c3_rocprof.log 2>&1 || exit 1
cd $R
for s in 0 2 3; do timeout -k 10 300 python bench.py --search-steps $s --no-cpu-baseline > gpurun_out/final_c3_s$s.log 2>&1 || exit 1; done
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/final_c2.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/final_c4.log 2>&1 || exit 1
timeout -k 10 600 python bench.py --workload c5 --base-rows 1250000 --queries 12500 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final_c5_share.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --rerank gather --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_c3_gather.log 2>&1 || exit 1
bash tools/pmc_traffic.sh > gpurun_out/pmc.log 2>&1 || exit 1
rm -rf gpurun_out/pmc[0-9]*
bash tools/pmc_scan.sh > gpurun_out/pmcscan.log 2>&1 || exit 1
python3 tools/pmc_detail.py k_leafscan_f16 k_rerank_short > gpurun_out/pmc_detail.json
python3 tools/last_batch.py gpurun_out/kt/kt_kernel_trace.csv 8 > gpurun_out/final_c3_timeline.txt
for f in c3 c3_s0 c3_s2 c3_s3 c2 c4 c5_share c3_gather; do
  echo $f $(tail -1 gpurun_out/final_$f.log | grep -o "\"value\": [0-9.]*\|ms_per_step\": [0-9.]*\|recall_at_10\": [0-9.]*\|\"frac\": [0-9.]*")
done

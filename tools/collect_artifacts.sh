#!/bin/bash
# Everything profiles/ is refreshed from, on the GPU box (run from the repository root).  Stages, so that one call
# stays within the box's time limit:  tools/collect_artifacts.sh tests | bench | pmc <workload> | timeline
# Afterwards, back in the container: tools/store_artifacts.sh <round tag>
R=$PWD
json_line() { grep '^{"metric"' "$1" | tail -1 > "$2"; test -s "$2"; }
case "$1" in
tests)
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 400 --timeout-method thread --durations=8 > gpurun_out/final_tests.log 2>&1 || { tail -20 gpurun_out/final_tests.log; exit 1; }
  tail -3 gpurun_out/final_tests.log ;;
bench)
  timeout -k 10 500 python bench.py > gpurun_out/final_c3.log 2>&1 && json_line gpurun_out/final_c3.log gpurun_out/final_c3.json || exit 1
  for s in 0 2 3; do timeout -k 10 300 python bench.py --search-steps $s --no-cpu-baseline --large-table-rows 0 > gpurun_out/final_c3_s$s.log 2>&1 && json_line gpurun_out/final_c3_s$s.log gpurun_out/final_c3_s$s.json || exit 1; done
  timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/final_c2.log 2>&1 && json_line gpurun_out/final_c2.log gpurun_out/final_c2.json || exit 1
  timeout -k 10 300 python bench.py --workload c4 --no-cpu-baseline > gpurun_out/final_c4.log 2>&1 && json_line gpurun_out/final_c4.log gpurun_out/final_c4.json || exit 1
  timeout -k 10 600 python bench.py --workload c5 --base-rows 1250000 --queries 12500 --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/final_c5_share.log 2>&1 && json_line gpurun_out/final_c5_share.log gpurun_out/final_c5_share.json || exit 1
  for f in c3 c3_s0 c3_s2 c3_s3 c2 c4 c5_share; do
    python3 -c "
import json; d = json.load(open('gpurun_out/final_$f.json')); r = d['roofline']
print('$f', round(d['value']), 'q/s', round(d['ms_per_step'], 3), 'ms; host-to-host', round(d.get('value_host_to_host') or 0), '; scan frac', round(r['frac'], 3), 'traffic', r.get('traffic'), 'recall', d['config'].get('recall_at_10'))"
  done ;;
pmc)
  W=$2
  case $W in
    c5) A="--workload c5 --base-rows 1250000 --queries 12500" ;;
    *) A="--workload $W" ;;
  esac
  bash tools/pmc_workload.sh $W $A || exit 1 ;;
timeline)
  cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/kt
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 > $R/gpurun_out/final_c3_rocprof.log 2>&1 || exit 1
  cd $R
  python3 tools/last_batch.py $(find gpurun_out/kt -name 'kt_kernel_trace.csv' | head -1) 8 > gpurun_out/final_c3_timeline.txt
  cat gpurun_out/final_c3_timeline.txt ;;
*) echo "usage: $0 tests | bench | pmc <c2|c3|c4|c5> | timeline"; exit 2 ;;
esac

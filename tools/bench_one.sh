#!/bin/bash
# usage: bash tools/bench_one.sh <tag> <workload> [bench.py arguments] : one batch timing without the side measurements
T=$1; W=$2; shift; shift
mkdir -p gpurun_out/q
timeout -k 10 250 python bench.py --workload $W --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 --no-gather-check "$@" 2>/dev/null | grep '^{"metric"' > gpurun_out/q/${T}_$W.json || exit 1
python3 -c "
import json
d = json.load(open('gpurun_out/q/${T}_$W.json')); r = d['roofline']
print('$T $W: batch', round(d['ms_per_step'], 3), 'ms, scan launch', round(r['ms_per_launch'], 3), 'ms x', r['launches_per_batch'], 'fwd/q', round(d['config']['forwarded_to_exact_rerank_per_query_on_rank0'], 1))"

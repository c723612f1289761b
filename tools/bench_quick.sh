#!/bin/bash
# usage: bash tools/bench_quick.sh <tag> [bench.py arguments] : c2, c3 and c4 batches without the side measurements, one line each
T=$1; shift
mkdir -p gpurun_out/q
for w in c2 c3 c4; do
  timeout -k 10 250 python bench.py --workload $w --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 "$@" 2>/dev/null | grep '^{"metric"' > gpurun_out/q/${T}_$w.json || exit 1
  python3 -c "
import json
d = json.load(open('gpurun_out/q/${T}_$w.json')); r = d['roofline']
print('$T $w: batch', round(d['ms_per_step'], 3), 'ms, scan launch', round(r['ms_per_launch'], 3), 'ms x', r['launches_per_batch'], 'fwd/q', round(d['config']['forwarded_to_exact_rerank_per_query_on_rank0'], 1))"
done

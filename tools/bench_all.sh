# usage: bash tools/bench_all.sh <tag> <workloads...> : one short bench per workload, key numbers on one line each
tag=$1; shift
for w in "$@"; do
  timeout -k 10 250 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 2>&1 | grep '^{"metric"' > gpurun_out/${tag}_$w.json || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/${tag}_$w.json')); r=d['roofline']
print('$tag $w: batch', round(d['ms_per_step'],3), 'ms, scan launch', round(r['ms_per_launch'],3), 'ms, frac', round(r['frac'],3), 'fwd/q', d['config'].get('forwarded_to_exact_rerank_per_query_on_rank0'))"
done

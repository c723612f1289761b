# usage: bash tools/bench_opts.sh <tag> "<bench.py arguments>" ... : one short bench per argument string, key numbers on one line each
tag=$1; shift
i=0
for a in "$@"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py $a --steps 12 --warmup 4 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 2>&1 | grep '^{"metric"' > gpurun_out/${tag}_$i.json || exit 1
  python -c "
import json
d=json.load(open('gpurun_out/${tag}_$i.json')); r=d['roofline']
print('$a: batch', round(d['ms_per_step'],3), 'ms, scan launches', r['launches_per_batch'], 'x', round(r['ms_per_launch'],3), 'ms, fwd/q', d['config'].get('forwarded_to_exact_rerank_per_query_on_rank0'))"
done

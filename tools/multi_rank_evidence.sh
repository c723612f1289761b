#!/bin/bash
# usage (GPU box, repository root): bash tools/multi_rank_evidence.sh <tag>
# What can be measured of the multi-GPU path on a one-GPU box, written under gpurun_out/<tag>/ for profiles/:
#  * per-rank batch time of every grid position's largest shard (collectives excluded), tools/shard_timing.py;
#  * bench.py with 2 and 4 ranks sharing the one GPU over gloo (--verify: merged result == unsharded index).
T=${1:-mr}
O=gpurun_out/$T
mkdir -p $O
: > $O/shard_timing.txt
for cfg in "tables 1 c3 1" "tables 2 c3 1" "tables 4 c3 1" "tables 4 c3 2" "tables 8 c3 1" "rows 2 c3 1" "rows 2 c3 2" "rows 2 c3 4" "rows 4 c3 2" "tables 4 c4 2" "rows 2 c4 4"; do
  timeout -k 10 200 python tools/shard_timing.py $cfg 2>/dev/null | grep "ms per batch" >> $O/shard_timing.txt || exit 1
done
cat $O/shard_timing.txt
for n in 2 4; do
  for mode in tables rows; do
    timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
      bench.py --gpus $n --backend gloo --mode $mode --verify --steps 10 --warmup 4 --no-recall > $O/gloo_${n}rank_${mode}_c3.json 2> $O/gloo_${n}rank_${mode}_c3.err || { tail -5 $O/gloo_${n}rank_${mode}_c3.err; exit 1; }
    grep -h "verify ok" $O/gloo_${n}rank_${mode}_c3.err
    python3 -c "
import json,sys
d=json.loads([l for l in open('$O/gloo_${n}rank_${mode}_c3.json') if l.startswith('{')][-1])
print('$n ranks $mode (gloo, one GPU shared):', round(d['ms_per_step'],3), 'ms per batch,', d['config']['parallelism'])"
  done
done

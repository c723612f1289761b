#!/bin/bash
# usage: bash tools/ablate_c5.sh <bits...> : the c5 share (1.25M base / 12.5k queries) on the ablation builds of the scan
for b in "$@"; do
  timeout -k 10 300 python tools/run_with_lib.py tools/bin/libdpf_ablate$b.so --workload c5 --base-rows 1250000 --queries 12500 --steps 6 --warmup 2 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 --no-gather-check 2>&1 | grep '^{"metric"' > gpurun_out/abl${b}_c5.json || exit 1
  python3 -c "
import json
d=json.load(open('gpurun_out/abl${b}_c5.json')); print('c5 share ablate$b: batch', round(d['ms_per_step'],3), 'ms, scan launch', round(d['roofline']['ms_per_launch'],3), 'ms x', d['roofline']['launches_per_batch'])"
done

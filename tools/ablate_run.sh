# usage: bash tools/ablate_run.sh <bits...> -- <workloads...> : batch and scan launch time of the measurement builds
# tools/bin/libdpf_ablate<bits>.so (made in the container by: make -C similaritysearchbyrdf_amd/csrc ablate BITS="1 3";
# bits: 1 loads and products only, 2 every query row from one hot spot, 4 no in-tile tightening)
set -e
bits=(); while [ "$1" != "--" ]; do bits+=("$1"); shift; done; shift
for w in "$@"; do for b in "${bits[@]}"; do
  timeout -k 10 250 python tools/run_with_lib.py tools/bin/libdpf_ablate$b.so --workload $w --steps 12 --warmup 4 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 2>&1 | grep '^{"metric"' > gpurun_out/abl${b}_$w.json
  python -c "
import json
d=json.load(open('gpurun_out/abl${b}_$w.json')); print('$w ablate$b: batch', round(d['ms_per_step'],3), 'ms, scan launch', round(d['roofline']['ms_per_launch'],3), 'ms x', d['roofline']['launches_per_batch'])"
done; done

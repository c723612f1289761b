#!/bin/bash
# Copies what tools/collect_artifacts.sh left under gpurun_out/ into profiles/ (argument: round tag, e.g. r02);
# stages that were not run are skipped.
T=${1:-r02}
cpif() { test -s "$1" && cp "$1" "$2"; }
cpif gpurun_out/final_tests.log profiles/${T}_gpu_tests.log
for f in c3 c3_s0 c3_s2 c3_s3 c2 c4 c5_share; do cpif gpurun_out/final_$f.json profiles/${T}_${f}_bench.json; done
cpif gpurun_out/final_c3_timeline.txt profiles/${T}_c3_batch_timeline.txt
for W in c2 c3 c4 c5; do
  test -d gpurun_out/pmc_$W/fetch || continue
  case $W in c5) N=1250000; Q=12500 ;; *) N=1000000; Q=10000 ;; esac
  case $W in c2) S=0 ;; *) S=1 ;; esac
  K=k_leafscan_f16; test $W = c4 && K=k_leafscan_i8     # (integer-valued rows: the int8 form of the scan runs)
  python3 tools/pmc_traffic.py $W $K $W $N $Q $S > profiles/${T}_${W}_${K}_pmc_traffic.json
  cpif "$(find gpurun_out/pmc_$W/kt -name 'kt_kernel_stats.csv' | head -1)" profiles/${T}_${W}_kernel_stats.csv
  grep '^{"metric"' gpurun_out/pmc_$W/kt.log | tail -1 > profiles/${T}_${W}_bench_under_rocprof.json
done
ls profiles | grep "^${T}_"

#!/bin/bash
# Copies what tools/collect_artifacts.sh left under gpurun_out/ into profiles/ (argument: round tag, e.g. r01)
T=${1:-r01}
cp gpurun_out/final_c3.log profiles/${T}_c3_bucket_major_bench.log
cp gpurun_out/kt/kt_kernel_stats.csv profiles/${T}_c3_bucket_major_kernel_stats.csv
cp gpurun_out/final_c2.log profiles/${T}_c2_bucket_major_bench.log
cp gpurun_out/final_c4.log profiles/${T}_c4_bucket_major_bench.log
cp gpurun_out/final_c5_share.log profiles/${T}_c5_share_bench.log
cp gpurun_out/final_c3_gather.log profiles/${T}_c3_gather_bench.log
for s in 0 2 3; do cp gpurun_out/final_c3_s$s.log profiles/${T}_c3_steps${s}_bench.log; done
cp gpurun_out/final_tests.log profiles/${T}_gpu_tests.log
cp gpurun_out/pmc_detail.json profiles/${T}_c3_pmc_detail.json
cp gpurun_out/final_c3_timeline.txt profiles/${T}_c3_batch_timeline.txt
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv k_leafscan_f16 c3 1 2 > profiles/${T}_c3_k_leafscan_f16_pmc_traffic.json
cp gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv profiles/${T}_c3_pmc_fetch_size.csv
cp gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv profiles/${T}_c3_pmc_write_size.csv

"""Averages per launch of the counters tools/pmc_scan.sh collected (gpurun_out/pmc1 .. pmcN), for the
kernels named on the command line: python tools/pmc_detail.py <workload label> k_leafscan_f16 k_rerank_short > profiles/..."""
import csv
import glob
import json
import sys
from collections import defaultdict

label = sys.argv[1] if len(sys.argv) > 1 else "c3"
want = sys.argv[2:] or ["k_leafscan_f16"]
out = defaultdict(dict)
for path in sorted(glob.glob("gpurun_out/pmc[0-9]*/p_counter_collection.csv") + glob.glob("gpurun_out/pmc[0-9]*/*/p_counter_collection.csv")):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if any(w in name for w in want):
                tot[(name, r["Counter_Name"])] += float(r["Counter_Value"])
                cnt[(name, r["Counter_Name"])] += 1
    for (name, c), v in tot.items():
        out[name][c] = v / cnt[(name, c)]
print(json.dumps({"workload": label, "what": "average per launch, one rocprofv3 --pmc pass per counter group (tools/pmc_scan.sh), "
                  "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-recall --no-host-to-host --large-table-rows 0 [--workload ...]", "kernels": out}, indent=1))

"""Turns the two counter_collection.csv files of tools/pmc_traffic.sh into the per-launch traffic
record bench.py reads: python tools/pmc_traffic.py <fetch.csv> <write.csv> <kernel substring>
<workload> <search steps> <launches per batch> > profiles/rNN_<workload>_<kernel>_pmc_traffic.json"""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            tot[name] += float(r["Counter_Value"])
            cnt[name] += 1
    return {n: {"launches": cnt[n], "avg_kb": tot[n] / cnt[n]} for n in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
key, workload, steps, launches = sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
name = [n for n in fetch if key in n][0]
f_raw = fetch[name]["avg_kb"] * 1024.0
w = write[name]["avg_kb"] * 1024.0
print(json.dumps({
    "workload": workload,
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 "
               "--no-cpu-baseline --no-recall (tools/pmc_traffic.sh)",
    "kernel": name, "launches_per_batch": launches, "search_steps": steps,
    "fetch_size_bytes_raw_per_launch": f_raw, "write_size_bytes_per_launch": w,
    "correction": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B: read bytes = 2 x FETCH_SIZE. Calibrated on this "
                  "kernel's own access shape (tools/pmc_calibrate.sh: 4.29 GB of 256-B rows pulled as MFMA operands "
                  "report FETCH_SIZE 2.1 GB and 34.1 M requests of 128 B).",
    "traffic_bytes_per_launch": 2.0 * f_raw + w,
    "per_kernel": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}, indent=1))

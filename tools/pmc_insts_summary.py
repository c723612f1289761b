"""Per-launch averages of the counters tools/pmc_insts.sh collected (gpurun_out/pmc_i1 .. pmc_i3) for the kernels whose
names contain one of the arguments, with the shares that say what a kernel's waves spend their cycles on:
python3 tools/pmc_insts_summary.py k_probe_lanes k_seed_radius"""
import csv
import glob
import json
import sys
from collections import defaultdict

want = sys.argv[1:] or ["k_probe_lanes"]
tot, cnt = defaultdict(float), defaultdict(int)
for path in sorted(glob.glob("gpurun_out/pmc_i[0-9]*/**/p_counter_collection.csv", recursive=True)):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0]
            if any(w in name for w in want):
                tot[(name, r["Counter_Name"])] += float(r["Counter_Value"])
                cnt[(name, r["Counter_Name"])] += 1
out = defaultdict(dict)
for (name, c), v in tot.items():
    out[name][c] = v / cnt[(name, c)]
for name, d in out.items():
    d["launches_seen"] = max(cnt[(name, c)] for c in d if (name, c) in cnt)
    if d.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if c in d:
                d["share_of_wave_cycles:" + c] = round(d[c] / d["SQ_WAVE_CYCLES"], 4)
    if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
        d["valu_active_per_busy_cycle"] = round(d["SQ_ACTIVE_INST_VALU"] / d["SQ_BUSY_CYCLES"], 4)
print(json.dumps(out, indent=1))

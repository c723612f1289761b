"""Turns the passes of tools/pmc_workload.sh into the per-launch record bench.py reads:
python tools/pmc_traffic.py <tag> <kernel substring> <workload> <base rows> <queries> <search steps>
    > profiles/rNN_<workload>_<kernel>_pmc_traffic.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

FP16_DENSE_PEAK_TFLOPS = 2500.0      # MI355X dense fp16 matrix peak
SIMDS = 256 * 4
CLOCK_GHZ = 2.4


def find(tag, sub, name):
    hits = glob.glob("gpurun_out/pmc_%s/%s/**/%s" % (tag, sub, name), recursive=True)
    if not hits:
        raise SystemExit("no %s under gpurun_out/pmc_%s/%s" % (name, tag, sub))
    return hits[0]


def per_kernel(path):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            tot[key] += float(r["Counter_Value"])
            cnt[key] += 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def kernel_stats(path):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            out[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                            "total_ns": float(r["TotalDurationNs"]), "percent": float(r["Percentage"])}
    return out


tag, key, workload, n, nq, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
fetch = per_kernel(find(tag, "fetch", "p_counter_collection.csv"))
write = per_kernel(find(tag, "write", "p_counter_collection.csv"))
mfma = per_kernel(find(tag, "mfma", "p_counter_collection.csv"))
stats = kernel_stats(find(tag, "kt", "kt_kernel_stats.csv"))
name = [k[0] for k in fetch if key in k[0]][0]
f_raw = fetch[(name, "FETCH_SIZE")][0] * 1024.0
w = write[(name, "WRITE_SIZE")][0] * 1024.0
avg_ns = stats[name]["avg_ns"]
rec = {
    "workload": workload, "base_rows": n, "queries": nq, "search_steps": steps,
    "command": "tools/pmc_workload.sh %s ...: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | matrix-core counters (separate passes) and "
               "--kernel-trace --stats -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-recall "
               "--no-host-to-host --large-table-rows 0 <workload arguments>" % tag,
    "kernel": name, "launches_profiled": fetch[(name, "FETCH_SIZE")][1],
    "avg_duration_us_kernel_trace": avg_ns / 1e3,
    "fetch_size_bytes_raw_per_launch": f_raw, "write_size_bytes_per_launch": w,
    "correction": "gfx950 FETCH_SIZE tallies 128-B requests at 64 B: read bytes = 2 x FETCH_SIZE (calibrated on this kernel's "
                  "own access shape, tools/pmc_calibrate.sh); both counters sit at the L2 <-> fabric boundary, so reads "
                  "served by the Infinity Cache are included",
    "traffic_bytes_per_launch": 2.0 * f_raw + w,
}
INT8_DENSE_PEAK_TOPS = 5000.0        # MI355X dense int8 matrix peak
mops = mfma.get((name, "SQ_INSTS_VALU_MFMA_MOPS_F16"))
mops8 = mfma.get((name, "SQ_INSTS_VALU_MFMA_MOPS_I8"))
if mops8 and mops8[0] > 0 and not (mops and mops[0] > 0):
    ops = mops8[0] * 512.0
    busy = mfma.get((name, "SQ_VALU_MFMA_BUSY_CYCLES"), (0.0, 0))[0]
    rec.update({
        "mfma_flops_per_launch": ops,
        "mfma_instructions_per_launch": mfma.get((name, "SQ_INSTS_MFMA"), (0.0, 0))[0],
        "mfma_busy_cycles_per_launch": busy,
        "mfma_tflops": ops / avg_ns / 1e3,
        "mfma_util": ops / avg_ns / 1e3 / INT8_DENSE_PEAK_TOPS,
        "mfma_busy_share_of_simd_cycles": busy / (avg_ns * CLOCK_GHZ * SIMDS),
        "mfma_util_definition": "SQ_INSTS_VALU_MFMA_MOPS_I8 x 512 op / kernel-trace duration / 5000 TOP/s (dense int8 peak); "
                                "busy share = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)",
    })
elif mops:
    flops = mops[0] * 512.0
    busy = mfma.get((name, "SQ_VALU_MFMA_BUSY_CYCLES"), (0.0, 0))[0]
    rec.update({
        "mfma_flops_per_launch": flops,
        "mfma_instructions_per_launch": mfma.get((name, "SQ_INSTS_MFMA"), (0.0, 0))[0],
        "mfma_busy_cycles_per_launch": busy,
        "mfma_tflops": flops / avg_ns / 1e3,
        "mfma_util": flops / avg_ns / 1e3 / FP16_DENSE_PEAK_TFLOPS,
        "mfma_busy_share_of_simd_cycles": busy / (avg_ns * CLOCK_GHZ * SIMDS),
        "mfma_util_definition": "SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 flop / kernel-trace duration / 2500 TFLOP/s (dense fp16 peak); "
                                "busy share = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs)",
    })
rec["kernel_stats"] = {k: v for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"])[:24]}
print(json.dumps(rec, indent=1))

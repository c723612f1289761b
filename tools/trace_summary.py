"""Per-kernel average duration from a rocprofv3 --kernel-trace CSV (argument: the *_kernel_trace.csv)."""
import csv
import sys
from collections import defaultdict

tot = defaultdict(float)
cnt = defaultdict(int)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0]
        name = name.replace("void dpf::", "").replace("dpf::", "")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot[name] += d
        cnt[name] += 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for name in sorted(tot, key=lambda n: -tot[n]):
    print("%-70s calls %6d  avg %9.1f us  per step %9.1f us" % (name[:70], cnt[name], tot[name] / cnt[name], tot[name] / steps))
print("sum per step: %.1f us" % (sum(tot.values()) / steps))

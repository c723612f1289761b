"""Durations of the scan launches in a rocprofv3 kernel trace: python tools/scan_passes.py <trace.csv>"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "leafscan" in r["Kernel_Name"]]
p1, p2 = d[0::2], d[1::2]
print("launches %d; first pass: median %.0f max %.0f us; second pass: median %.0f max %.0f us" % (
    len(d), sorted(p1)[len(p1) // 2], max(p1), sorted(p2)[len(p2) // 2], max(p2)))

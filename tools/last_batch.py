"""Timeline of the last bucket-major batch in a rocprofv3 kernel trace: python tools/last_batch.py <trace.csv> [n-th last scan launch]"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ls = [i for i, r in enumerate(rows) if "k_leafscan" in r["Kernel_Name"]]
i0 = ls[-back]
while "k_hash" not in rows[i0]["Kernel_Name"]:
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
prev = t0
i = i0
while True:
    r = rows[i]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f gap %6.1f dur %7.1f %s" % ((st - t0) / 1e3, (st - prev) / 1e3, (en - st) / 1e3,
                                             r["Kernel_Name"].split("(")[0].replace("void dpf::", "")[:40]))
    prev = en
    i += 1
    if i >= len(rows) or "k_hash" in rows[i]["Kernel_Name"] or "copyBuffer" in rows[i]["Kernel_Name"]:
        break                     # the next batch, or the caller's copies behind the last one

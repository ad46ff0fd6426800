"""Summarise the tail of a rocprofv3 kernel trace (p_kernel_trace.csv): the last N dispatches with duration and the gap
to the previous one.  python scripts/trace_gaps.py <csv> [n_last]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tail = rows[-n:]
prev = None
tot_k = tot_g = 0
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]
    print(f"{name:46s} {((e - s) / 1e3):8.1f} us  gap {gap:8.1f} us")
    tot_k += e - s
    tot_g += max(gap, 0) * 1e3
    prev = e
print("kernels %.1f us, gaps %.1f us" % (tot_k / 1e3, tot_g / 1e3))

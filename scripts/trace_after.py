"""Per-kernel totals of a rocprofv3 kernel trace (p_kernel_trace.csv) from the first dispatch whose name contains
<substring> on: python scripts/trace_after.py <csv> <substring> [top]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = min(int(r["Start_Timestamp"]) for r in rows if sys.argv[2] in r["Kernel_Name"])
agg = collections.defaultdict(lambda: [0, 0])
first, last = None, 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0:
        continue
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    agg[name][0] += 1
    agg[name][1] += e - s
    first = s if first is None else first
    last = max(last, e)
tot = sum(v[1] for v in agg.values())
print(f"# from the first {sys.argv[2]}: kernels busy {tot / 1e6:.2f} ms of {(last - first) / 1e6:.2f} ms wall")
print("kernel,calls,total_ms,avg_us,percent")
for name, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{name},{v[0]},{v[1] / 1e6:.3f},{v[1] / v[0] / 1e3:.1f},{100 * v[1] / tot:.1f}")

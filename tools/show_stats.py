#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv: top kernels, total GPU time."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU kernel time {tot/1e6:.2f} ms over {int(steps)} steps = {tot/1e6/steps:.2f} ms/step; {sum(int(r['Calls']) for r in rows)/steps:.0f} launches/step")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:90]
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{float(r['TotalDurationNs'])/1e6/steps:8.3f} ms/step  {int(r['Calls'])/steps:7.1f} calls/step  avg {float(r['AverageNs'])/1e3:8.1f} us  {short(r['Name'])}")

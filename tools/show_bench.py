#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line (kernel table sorted by time)."""
import json, sys
d = json.load(open(sys.argv[1]))
k = d.pop("kernels", {})
print(json.dumps({a: d[a] for a in ("metric", "value", "ms_per_step", "dtype", "n_gpus")}))
print("roofline:", json.dumps(d.get("roofline")))
print("cpu_baseline:", json.dumps(d.get("cpu_baseline")))
tot = 0
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += v["ms_per_step"]
    print(f"  {n:30s} n/step {v['launches_per_step']:6.1f} ms/step {v['ms_per_step']:7.3f} TF/s {v['TFLOPs'] or 0:7.1f} GB/s {v['GBs']:7.1f}")
print("  profiled kernels ms/step:", round(tot, 3))

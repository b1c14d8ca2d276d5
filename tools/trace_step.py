#!/usr/bin/env python3
"""Per-kernel breakdown of the LAST full training step in a rocprofv3 kernel trace (csv), delimited by pack_multi_kernel
launches (the first kernel of a step).  Usage: tools/trace_step.py <kernel_trace.csv> [top_n]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_multi' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
step = rows[a:b]
t0, t1 = int(step[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return n[:100]
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[short(r['Kernel_Name'])][0] += 1
    agg[short(r['Kernel_Name'])][1] += d
    busy += d
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels, sum of kernel durations {busy / 1e6:.3f} ms")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{d / 1e6:7.3f} ms {n:4d}x avg {d / n / 1e3:7.1f} us  {k}")
if len(sys.argv) > 3:                      # ordered dump of the step: index, start offset, duration, grid/block, name
    with open(sys.argv[3], "w") as f:
        for i, r in enumerate(step):
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            f.write(f"{i:4d} +{(s - t0) / 1e3:8.1f} us {(e - s) / 1e3:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4} lds {r.get('LDS_Block_Size', '?'):>6} vgpr {r.get('VGPR_Count', '?'):>4}  {short(r['Kernel_Name'])}\n")

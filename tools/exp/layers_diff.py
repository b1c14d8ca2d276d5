"""Difference of two bench.py --dump-layers tables (us per step, per (layer, direction)).  usage: layers_diff.py A.txt B.txt [n]"""
import re, sys
def load(p):
    d = {}
    for l in open(p).read().splitlines()[1:]:
        m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(.*)', l)
        if m:
            lab = m.group(6); lay = lab.split('|')[1].strip()
            k = (lay + ' ' + ('dgrad' if '/dgrad' in lab else 'fwd' if '/fwd' in lab else '')) if lay else lab
            d.setdefault(k, [0, 0, lab]); d[k][0] += float(m.group(1)); d[k][1] += float(m.group(2))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
print("total us/step", round(sum(v[0] for v in a.values()), 1), "->", round(sum(v[0] for v in b.values()), 1))
rows = sorted((b.get(k, [0])[0] - a.get(k, [0])[0], k, a.get(k, [0, 0, ''])[0], b.get(k, [0, 0, ''])[0], (b.get(k) or a.get(k))[2].split('|')[0].strip(), (b.get(k) or a.get(k))[1]) for k in set(a) | set(b))
for r in rows[:n] + [None] + rows[-n:]:
    print('...' if r is None else f"{r[0]:8.1f} {r[2]:8.1f} -> {r[3]:8.1f}  x{r[5]:.0f}  {r[1]}   [{r[4]}]")

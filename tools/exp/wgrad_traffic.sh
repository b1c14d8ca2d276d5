#!/bin/bash
# HBM-side bytes of the grouped weight-gradient launches, per kernel symbol (one FETCH_SIZE and one WRITE_SIZE pass over eager steps)
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/wgt_$c
  timeout -k 10 300 rocprofv3 -M --pmc $c --kernel-trace --output-format csv -d gpurun_out/wgt_$c -o p -- python3 bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-also "$@" > gpurun_out/wgt_$c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/wgt_{c}/**/p_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c or "wgrad" not in r["Kernel_Name"]:
            continue
        agg[r["Kernel_Name"][:60]][0] += 1
        agg[r["Kernel_Name"][:60]][1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        out.setdefault(k, {})[c] = (2.0 if c == "FETCH_SIZE" else 1.0) * 1024.0 * v / n
for k, v in out.items():
    print(f"{k:62s} fetch {v.get('FETCH_SIZE', 0) / 1e6:8.1f} MB  write {v.get('WRITE_SIZE', 0) / 1e6:8.1f} MB per launch")
PY
rm -rf gpurun_out/wgt_FETCH_SIZE gpurun_out/wgt_WRITE_SIZE

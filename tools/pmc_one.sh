#!/bin/bash
# SQ counters of ONE layer of tools/bench_ops.py (call through gpurun).  usage: tools/pmc_one.sh <tag> <set: s|m> <mode fwd|dgrad|wgrad> <layer substring>
set -e
tag=$1; export DSN_BENCH_SET=$2; mode=$3; shift 3
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$out" -o $tag -- python3 tools/bench_ops.py $mode "$@" > gpurun_out/pmc_$tag.log 2>&1
f=$(find "$out" -name "${tag}_counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in agg.items():
    if "igemm" not in k and "wgrad" not in k: continue
    n = max(cnt[k], 1); wc = v["SQ_WAVE_CYCLES"] or 1
    print(k, "launches", n)
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x / n:14.0f} per launch   {x / wc:6.3f} of WAVE_CYCLES")
PY

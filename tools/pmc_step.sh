#!/bin/bash
# SQ counters of every kernel of a training step (call through gpurun): one rocprofv3 --pmc pass over a few eager steps, folded per
# kernel symbol.  usage: tools/pmc_step.sh <tag> [bench args]   ->  gpurun_out/pmc_<tag>_summary.txt
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/pmcstep_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d "$out" -o p -- python3 bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-also "$@" > gpurun_out/pmcstep_$tag.log 2>&1 || { tail -5 gpurun_out/pmcstep_$tag.log; exit 1; }
f=$(find "$out" -name "p_counter_collection.csv" | head -1)
python3 - "$f" > gpurun_out/pmc_${tag}_summary.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    k = re.sub(r"^void \(anonymous namespace\)::", "", k)[:86]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
rows = []
for k, v in agg.items():
    n = max(cnt[k], 1); wc = v["SQ_WAVE_CYCLES"] or 1.0; busy = v["SQ_BUSY_CYCLES"] or 1.0
    rows.append((v["SQ_BUSY_CYCLES"], k, n, v["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_ACTIVE_INST_LDS"], 1.0),
                 v["SQ_WAIT_INST_ANY"] / wc, v["SQ_INSTS_VALU"] / max(v["SQ_INSTS_MFMA"], 1.0)))
print(f"{'kernel':88s} {'launches':>8s} {'MFMA busy / SQ busy':>20s} {'LDS conflict / LDS active':>26s} {'wait / wave cycles':>19s} {'VALU per MFMA':>14s}")
for _, k, n, mf, lc, wt, vm in sorted(rows, reverse=True)[:40]:
    print(f"{k:88s} {n:8d} {mf:20.3f} {lc:26.3f} {wt:19.3f} {vm:14.1f}")
PY
rm -rf "$out"
cat gpurun_out/pmc_${tag}_summary.txt

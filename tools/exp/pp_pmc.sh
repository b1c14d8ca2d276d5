#!/bin/bash
# SQ / GRBM counters of the big-tile 3x3 kernel on one bench_ops layer.  usage (through gpurun): tools/exp/pp_pmc.sh <tag> <DSN_PP mode> <set> <layer substring>
set -e
tag=$1; export DSN_PP=$2; export DSN_BENCH_SET=$3; shift 3
export TMPDIR=/tmp
for pass in a b; do
  out=gpurun_out/pmc_${tag}_$pass
  rm -rf "$out"
  if [ $pass = a ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE";
  else C="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE"; fi
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$out" -o $tag -- python3 tools/bench_ops.py ${PMC_WHICH:-fwd} "$@" > gpurun_out/pmc_${tag}_$pass.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_$pass.log; }
  f=$(find "$out" -name "${tag}_counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in agg.items():
    if "conv3x3" not in k and "wgrad_pp" not in k: continue
    n = max(cnt[k], 1); wc = v["SQ_WAVE_CYCLES"] or 1
    print(k, "launches", n)
    for c, x in sorted(v.items()):
        print(f"   {c:32s} {x / n:16.0f} per launch   {x / wc:8.4f} of WAVE_CYCLES")
PY
done

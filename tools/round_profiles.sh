#!/bin/bash
# Everything a round's profiles/ needs, in one gpurun call: rocprofv3 kernel-trace summaries of the three benchmark configurations
# and the two PMC passes behind profiles/traffic.json.  usage: tools/round_profiles.sh <tag, e.g. r02j>
tag=$1
export TMPDIR=/tmp
timeout -k 10 300 bash tools/prof.sh ${tag}_train_bs8_bf16 --steps 20 --warmup 5 || exit 1
timeout -k 10 300 bash tools/prof.sh ${tag}_infer_bs16_fp32 --mode infer --dtype fp32 --batch 16 --steps 20 --warmup 5 || exit 1
timeout -k 10 400 bash tools/prof.sh ${tag}_cfg5_m1280_bs4_bf16 --model m --img 1280 --batch 4 --steps 8 --warmup 3 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  timeout -k 10 400 rocprofv3 -M --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -o p -- python3 bench.py --eager --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-also > gpurun_out/pmc_${tag}_$c.log 2>&1 || exit 1
done
f=$(find gpurun_out/pmc_${tag}_FETCH_SIZE -name "p_counter_collection.csv" | head -1)
w=$(find gpurun_out/pmc_${tag}_WRITE_SIZE -name "p_counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$f" "$w" > gpurun_out/traffic_${tag}.json && cat gpurun_out/traffic_${tag}.json
# (the raw counter CSVs are large: only the summary travels back)
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE

for cfg in "B=256 M=1024" "B=192 M=1536" "B=128 M=2048" "B=256 M=2048" "B=128 M=1024"; do
  eval $cfg
  echo "== blocks=$B minpx=$M"
  DSN_WGRAD_BLOCKS=$B DSN_WGRAD_MINPX=$M timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null > gpurun_out/bs.json
  python tools/show_bench.py gpurun_out/bs.json > gpurun_out/bs.txt; grep -E "value|wgrad" gpurun_out/bs.txt
done

for cfg in "A=1 B=512 M=512" "A=1 B=256 M=512" "A=1 B=1024 M=256" "A=1 B=128 M=1024"; do
  eval $cfg
  echo "== alltaps=$A blocks=$B minpx=$M"
  DSN_WGRAD_BLOCKS=$B DSN_WGRAD_MINPX=$M DSN_WGRAD_ALLTAPS=$A timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null > gpurun_out/bs.json
  python tools/show_bench.py gpurun_out/bs.json | grep -E "value|wgrad"
done

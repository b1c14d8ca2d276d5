for cfg in "G=32 B=512 M=512" "G=64 B=512 M=512" "G=64 B=256 M=1024" "G=32 B=256 M=1024" "G=64 B=128 M=2048"; do
  eval $cfg
  echo "== gpk=$G blocks=$B minpx=$M"
  DSN_WGRAD_GPK=$G DSN_WGRAD_BLOCKS=$B DSN_WGRAD_MINPX=$M timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 4 2>/dev/null > gpurun_out/bs.json
  python tools/show_bench.py gpurun_out/bs.json | grep -E "value|wgrad"
done

for cfg in "A=1 B=1024 C=256" "A=0 B=1024 C=256" "A=1 B=1024 C=1024" "A=1 B=512 C=512" "A=0 B=512 C=256" "A=1 B=2048 C=2048"; do
  eval $cfg
  echo "== alltaps=$A blocks=$B scap=$C"
  DSN_WGRAD_ALLTAPS=$A DSN_WGRAD_BLOCKS=$B DSN_WGRAD_SCAP=$C python tools/bench_ops.py wgrad 2>&1 | grep -v amdgpu.ids
done

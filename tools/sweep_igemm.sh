# per-layer tile sweep: default heuristic (cfg -1) vs every forced tile; output: gpurun_out/sweep_<mode>_<cfg>.txt
for mode in fwd dgrad; do
  for c in -1 0 1 2 3 5 6; do
    if [ "$c" = "-1" ]; then timeout -k 10 120 python tools/bench_ops.py $mode > gpurun_out/sweep_${mode}_def.txt 2>&1;
    else DSN_IGEMM_CFG=$c timeout -k 10 120 python tools/bench_ops.py $mode > gpurun_out/sweep_${mode}_$c.txt 2>&1; fi
  done
done
echo sweep done

set -o pipefail
timeout -k 10 600 python -m pytest tests/test_pp_gpu.py -x -q -k "wgrad" > gpurun_out/wpp_test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/wpp_test.log
tail -12 gpurun_out/wpp_test.log
if grep -q "pytest rc=0" gpurun_out/wpp_test.log; then
  : > gpurun_out/wpp_bench.log
  for m in 0 1; do
    echo "== DSN_WGRAD_PP=$m config3" >> gpurun_out/wpp_bench.log
    DSN_WGRAD_PP=$m timeout -k 10 300 python tools/bench_ops.py wgrad "ffm" >> gpurun_out/wpp_bench.log 2>&1
    echo "== DSN_WGRAD_PP=$m config5" >> gpurun_out/wpp_bench.log
    DSN_WGRAD_PP=$m DSN_BENCH_SET=m timeout -k 10 300 python tools/bench_ops.py wgrad "k3 @" >> gpurun_out/wpp_bench.log 2>&1
  done
  grep -v amdgpu.ids gpurun_out/wpp_bench.log
fi

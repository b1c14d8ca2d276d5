set -o pipefail
timeout -k 10 600 python -m pytest tests/test_pp_gpu.py -x -q > gpurun_out/pp_test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pp_test.log
tail -5 gpurun_out/pp_test.log
if grep -q "pytest rc=0" gpurun_out/pp_test.log; then
  for m in ${PP_MODES:-2 3}; do
    echo "== DSN_PP=$m config3" >> gpurun_out/pp_bench.log
    DSN_PP=$m timeout -k 10 300 python tools/bench_ops.py ${PP_WHICH:-fwd} "ffm" >> gpurun_out/pp_bench.log 2>&1
    echo "== DSN_PP=$m config5" >> gpurun_out/pp_bench.log
    DSN_PP=$m DSN_BENCH_SET=m timeout -k 10 300 python tools/bench_ops.py ${PP_WHICH:-fwd} "k3 @" >> gpurun_out/pp_bench.log 2>&1
    echo "== DSN_PP=$m K scan" >> gpurun_out/pp_bench.log
    DSN_PP=$m DSN_BENCH_SET=pp timeout -k 10 300 python tools/bench_ops.py ${PP_WHICH:-fwd} >> gpurun_out/pp_bench.log 2>&1
  done
  grep -v amdgpu.ids gpurun_out/pp_bench.log
fi

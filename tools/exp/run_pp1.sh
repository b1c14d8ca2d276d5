set -o pipefail
timeout -k 10 600 python -m pytest tests/test_pp1_gpu.py -x -q > gpurun_out/pp1_test.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pp1_test.log
tail -5 gpurun_out/pp1_test.log
rm -f gpurun_out/pp1_bench.log
if grep -q "pytest rc=0" gpurun_out/pp1_test.log; then
  for m in ${PP1_MODES:-0 2 3}; do
    echo "== DSN_PP1=$m" >> gpurun_out/pp1_bench.log
    DSN_PP1=$m DSN_BENCH_SET=1x1 timeout -k 10 300 python tools/bench_ops.py ${PP_WHICH:-fwd} >> gpurun_out/pp1_bench.log 2>&1
    DSN_PP1=$m DSN_BENCH_SET=1x1 timeout -k 10 300 python tools/bench_ops.py dgrad >> gpurun_out/pp1_bench.log 2>&1
  done
  grep -v amdgpu.ids gpurun_out/pp1_bench.log
fi

set -o pipefail
rm -f gpurun_out/ppdir_bench.log
for d in 0 1; do
  echo "== DSN_PP_DIR=$d DSN_PP=5 (128-channel tiles, one block per CU)" >> gpurun_out/ppdir_bench.log
  DSN_PP_DIR=$d DSN_PP=5 timeout -k 10 300 python tools/bench_ops.py fwd "ffm" >> gpurun_out/ppdir_bench.log 2>&1
  DSN_PP_DIR=$d DSN_PP=5 DSN_BENCH_SET=m timeout -k 10 300 python tools/bench_ops.py fwd "k3 @" >> gpurun_out/ppdir_bench.log 2>&1
  DSN_PP_DIR=$d DSN_PP=5 DSN_BENCH_SET=m timeout -k 10 300 python tools/bench_ops.py dgrad "k3 @" >> gpurun_out/ppdir_bench.log 2>&1
  DSN_PP_DIR=$d DSN_PP=5 DSN_BENCH_SET=pp timeout -k 10 300 python tools/bench_ops.py fwd "pp " >> gpurun_out/ppdir_bench.log 2>&1
  DSN_PP_DIR=$d DSN_PP1=2 DSN_BENCH_SET=1x1 timeout -k 10 300 python tools/bench_ops.py fwd >> gpurun_out/ppdir_bench.log 2>&1
done
grep -v amdgpu.ids gpurun_out/ppdir_bench.log
bash tools/exp/run_ab.sh "DSN_PP_DIR=1" ab_ppdir

for u in 0 1 2 4; do echo "== DSN_EW_UNROLL=$u"; DSN_EW_UNROLL=$u python tools/bench_ops.py bn 2>&1 | grep -v amdgpu.ids; done

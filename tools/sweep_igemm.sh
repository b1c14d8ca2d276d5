for c in 2 5 6; do echo "== DSN_IGEMM_CFG=$c"; DSN_IGEMM_CFG=$c python tools/bench_ops.py all 2>&1 | grep -v amdgpu.ids | awk -F'|' '{print $1 "|" $2 "|" $3}' | sed 's/ *GF[^|]*|/|/' ; done

#!/bin/bash
# A/B of environment tuning knobs on a bench configuration (call through gpurun).
# usage: tools/sweep_env.sh "<bench args>" "VAR=a VAR=b ..." ...   -- one run per listed assignment set (use , to join several variables)
args=$1; shift
run() { env "$@" python3 bench.py $args --no-also --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"; }
echo "baseline $(run DSN_NOP=1) $(run DSN_NOP=1)"
for set in "$@"; do
  echo "$set $(run $(echo $set | tr ',' ' '))"
done

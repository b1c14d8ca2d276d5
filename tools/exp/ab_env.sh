# A/B of one environment setting on the whole step (config 3 / 5, alternated).  usage: ab_env.sh "VAR=value" [tag]
set -o pipefail
SETTING="$1"; tag=${2:-ab_env}; log=gpurun_out/${tag}.log; : > $log
for rep in 1 2 3; do
  for arm in A B; do
    if [ $arm = A ]; then pre=""; else pre="$SETTING"; fi
    echo "== $arm ($pre) config3 rep $rep" >> $log
    env $pre timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-also --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" >> $log
    echo "== $arm ($pre) config5 rep $rep" >> $log
    env $pre timeout -k 10 300 python bench.py --model m --img 1280 --batch 4 --steps 12 --warmup 4 --no-also --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" >> $log
  done
done
cat $log

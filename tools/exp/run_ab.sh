# A/B of one environment switch on the whole step: config 3 and config 5, alternated.  usage: run_ab.sh "<ENV=val for arm B>" [tag]
set -o pipefail
B="$1"; tag=${2:-ab}
log=gpurun_out/${tag}.log; : > $log
for rep in 1 2; do
  for arm in A B; do
    if [ $arm = A ]; then E=""; else E="$B"; fi
    echo "== $arm [$E] config3 rep $rep" >> $log
    env $E timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-also --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" >> $log
    echo "== $arm [$E] config5 rep $rep" >> $log
    env $E timeout -k 10 300 python bench.py --model m --img 1280 --batch 4 --steps 12 --warmup 4 --no-also --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" >> $log
  done
done
cat $log

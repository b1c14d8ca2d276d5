# per-layer in-step tables of config 5 (and 3) for two environment settings.  usage: layers_ab.sh "<ENV A>" "<ENV B>" tag
set -o pipefail
A="$1"; B="$2"; tag=${3:-lab}
for arm in A B; do
  if [ $arm = A ]; then E="$A"; else E="$B"; fi
  env $E timeout -k 10 300 python bench.py --model m --img 1280 --batch 4 --no-also --no-cpu-baseline --steps 10 --warmup 3 --dump-layers gpurun_out/${tag}_${arm}_cfg5_layers.txt > gpurun_out/${tag}_${arm}_c5.log 2>&1
  env $E timeout -k 10 300 python bench.py --no-also --no-cpu-baseline --dump-layers gpurun_out/${tag}_${arm}_cfg3_layers.txt > gpurun_out/${tag}_${arm}_c3.log 2>&1
  python -c "
import json,sys
for c in ('c5','c3'):
    j=json.loads(open('gpurun_out/${tag}_${arm}_'+c+'.log').read().strip().splitlines()[-1]); print('$arm [$E]', c, j['value'], j['ms_per_step'])
"
done

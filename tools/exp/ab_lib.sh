# A/B of two builds of the library on the whole step (config 3 / 5, alternated).  usage: ab_lib.sh <other .so> [tag]
set -o pipefail
OTHER="$1"; tag=${2:-ab_lib}; log=gpurun_out/${tag}.log; : > $log
cp desenet_amd/libdesenet_hip.so /tmp/lib_A.so; cp "$OTHER" /tmp/lib_B.so
for rep in 1 2 3; do
  for arm in A B; do
    cp /tmp/lib_$arm.so desenet_amd/libdesenet_hip.so
    echo "== $arm config3 rep $rep" >> $log
    timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-also --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" >> $log
    echo "== $arm config5 rep $rep" >> $log
    timeout -k 10 300 python bench.py --model m --img 1280 --batch 4 --steps 12 --warmup 4 --no-also --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" >> $log
  done
done
cp /tmp/lib_A.so desenet_amd/libdesenet_hip.so
cat $log

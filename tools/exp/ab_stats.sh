# per-kernel totals of config 3 with two builds of the library (call through gpurun from the repo root).  usage: ab_stats.sh <other .so> [tag]
set -o pipefail
OTHER="$1"; tag=${2:-ab_stats}
cp desenet_amd/libdesenet_hip.so /tmp/lib_A.so; cp "$OTHER" /tmp/lib_B.so
export TMPDIR=/tmp
for arm in A B; do
  cp /tmp/lib_$arm.so desenet_amd/libdesenet_hip.so
  bash tools/prof.sh ${tag}_$arm --steps 30 --warmup 8 || { cp /tmp/lib_A.so desenet_amd/libdesenet_hip.so; exit 1; }
done
cp /tmp/lib_A.so desenet_amd/libdesenet_hip.so

#!/bin/bash
# rocprofv3 kernel-trace summaries of bench.py runs on the GPU box (call through gpurun).  usage: tools/prof.sh <tag> <bench args...>
# writes gpurun_out/prof_<tag>/ (kernel_stats.csv, kernel_trace.csv) and gpurun_out/prof_<tag>_last_step.txt (train runs)
set -e
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
rm -rf "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 bench.py "$@" --no-cpu-baseline --no-profile --no-also > gpurun_out/prof_$tag.log 2>&1
f=$(find "$out" -name "${tag}_kernel_stats.csv" | head -1)
t=$(find "$out" -name "${tag}_kernel_trace.csv" | head -1)
echo "stats: $f"
grep -q pack_multi "$t" 2>/dev/null && python3 tools/trace_step.py "$t" 60 gpurun_out/prof_${tag}_ordered.txt > gpurun_out/prof_${tag}_last_step.txt || true
tail -2 gpurun_out/prof_$tag.log

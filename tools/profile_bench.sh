#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + stats of the bench, summaries under gpurun_out/.
# usage: tools/profile_bench.sh <tag> [bench args...]
set -e
tag=${1:-r01}; shift || true
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o trace -- \
  python3 bench.py --no-headline --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
find "$out" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
python3 tools/summarize_trace.py "$out" > "$out/summary.txt"
cat "$out/summary.txt"

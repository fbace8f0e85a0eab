#!/bin/bash
# usage: tools/profile_cmd.sh <tag> <python script> [args...]  -> gpurun_out/prof_<tag>/summary.txt
set -e
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o trace -- python3 "$@" > "$out/stdout.txt" 2> "$out/stderr.txt"
python3 tools/summarize_trace.py "$out" > "$out/summary.txt"
cat "$out/stdout.txt"
head -20 "$out/summary.txt"

#!/bin/bash
# instruction-cache / fetch counters for one shape. usage: tools/profile_icache.sh <tag> <edge> <nmaps> <algo>
set -e
tag=$1; shift
export TMPDIR=/tmp
rocprofv3 --list-avail > gpurun_out/avail_$tag.txt 2>&1 || true
grep -i -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_LEVEL\)[A-Z_]*" gpurun_out/avail_$tag.txt | sort -u > gpurun_out/avail_icache_$tag.txt || true
cat gpurun_out/avail_icache_$tag.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_IFETCH SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  out=gpurun_out/ic_${tag}_$i
  mkdir -p "$out"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -o pmc -- python3 tools/probe_one.py "$@" > "$out/stdout.txt" 2> "$out/stderr.txt" || { tail -5 "$out/stderr.txt"; }
  i=$((i+1))
done
python3 tools/sq_summary.py gpurun_out/ic_${tag}_0 gpurun_out/ic_${tag}_1

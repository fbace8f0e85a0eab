#!/bin/bash
# PMC passes (separate runs, kernel-trace only alongside): FETCH_SIZE then WRITE_SIZE.
set -e
tag=${1:-r01}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmc_${tag}_$c
  mkdir -p "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -o pmc -- python3 tools/pmc_probe.py > "$out/stdout.txt" 2> "$out/stderr.txt"
done
python3 tools/pmc_traffic.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE > gpurun_out/pmc_traffic_$tag.json
cat gpurun_out/pmc_traffic_$tag.json

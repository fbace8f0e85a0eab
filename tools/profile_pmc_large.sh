#!/bin/bash
# PMC passes for the mid-size / large-tile kernels (separate runs, kernel-trace only alongside): FETCH_SIZE then WRITE_SIZE.
#   tools/profile_pmc_large.sh <tag>   -> gpurun_out/pmc_traffic_large_<tag>.json (copy to profiles/pmc_traffic_large.json)
set -e
tag=${1:-r03}
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmcl_${tag}_$c
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -o pmc -- python3 tools/pmc_probe_large.py > "$out/stdout.txt" 2> "$out/stderr.txt"
done
python3 tools/pmc_traffic_large.py gpurun_out/pmcl_${tag}_FETCH_SIZE gpurun_out/pmcl_${tag}_WRITE_SIZE > gpurun_out/pmc_traffic_large_$tag.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_traffic_large_$tag.json"))
for k, v in d["kernels"].items():
    print("%-40s %4d  hbm/alg %.4f" % (k, v["edge"], v["hbm_over_alg"]))
PY

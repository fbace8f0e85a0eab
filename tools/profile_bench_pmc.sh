#!/bin/bash
# HBM traffic of the bench's own kernels at the bench's own launch sizes. Run on the GPU box (via gpurun):
#   tools/profile_bench_pmc.sh <tag> [bench args...]
# Two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, then WRITE_SIZE; --kernel-trace only alongside, as the
# guide prescribes), the read side calibrated in the SAME run against a read of known size in the codelet
# kernels' access width (bench.py --pmc-calib). Result: gpurun_out/pmc_traffic_bench_<tag>.json; copy it to
# profiles/pmc_traffic_bench.json, where bench.py picks up `roofline.traffic` for the dominant kernel.
set -e
tag=${1:-r02}; shift || true
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmcb_${tag}_$c
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -o pmc -- \
    python3 bench.py --no-headline --no-cpu-baseline --steps 5 --warmup 2 --pmc-calib "$@" > "$out/bench.json" 2> "$out/stderr.txt"
done
python3 tools/pmc_bench_traffic.py gpurun_out/pmcb_${tag}_FETCH_SIZE gpurun_out/pmcb_${tag}_WRITE_SIZE > gpurun_out/pmc_traffic_bench_$tag.json
cat gpurun_out/pmc_traffic_bench_$tag.json

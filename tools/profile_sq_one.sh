#!/bin/bash
# usage: tools/profile_sq_one.sh <tag> <edge> <nmaps> <algo>
set -e
tag=$1; shift
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  out=gpurun_out/sq1_${tag}_$i
  mkdir -p "$out"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -o pmc -- python3 tools/probe_one.py "$@" > "$out/stdout.txt" 2> "$out/stderr.txt" || { tail -5 "$out/stderr.txt"; }
  i=$((i+1))
done
python3 tools/sq_summary.py gpurun_out/sq1_${tag}_0 gpurun_out/sq1_${tag}_1 gpurun_out/sq1_${tag}_2

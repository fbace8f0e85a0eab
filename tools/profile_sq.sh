#!/bin/bash
# SQ counter passes for the probe workload (kernel-trace alongside only).
set -e
tag=${1:-r01}
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  out=gpurun_out/sq_${tag}_$i
  mkdir -p "$out"
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$out" -o pmc -- python3 tools/pmc_probe.py > "$out/stdout.txt" 2> "$out/stderr.txt" || { tail -5 "$out/stderr.txt"; }
  i=$((i+1))
done
python3 tools/sq_summary.py gpurun_out/sq_${tag}_0 gpurun_out/sq_${tag}_1 gpurun_out/sq_${tag}_2 > gpurun_out/sq_summary_$tag.txt
cat gpurun_out/sq_summary_$tag.txt

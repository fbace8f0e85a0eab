# round-3 evidence run B (GPU box): PMC traffic of the bench's own launches and of every large-tile kernel, the other nets' bench lines,
# kernel trace of the U2-Net-p step
export TMPDIR=/tmp
mkdir -p gpurun_out/fb
bash tools/profile_bench_pmc.sh r03 > gpurun_out/fb/pmc_bench.txt 2>&1 || { tail gpurun_out/fb/pmc_bench.txt; exit 1; }
bash tools/profile_pmc_large.sh r03b > gpurun_out/fb/pmc_large.txt 2>&1 || { tail gpurun_out/fb/pmc_large.txt; exit 1; }
tail -14 gpurun_out/fb/pmc_large.txt
: > gpurun_out/fb/other_nets.jsonl
for net in vgg_16_bn resnet_56 resnet_110 densenet_40 googlenet; do
  python bench.py --net $net --no-headline --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/fb/other_nets.jsonl
done
python bench.py --net u2netp --no-headline --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/fb/other_nets.jsonl
python bench.py --net u2netp --input-size 320 --no-headline --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/fb/other_nets.jsonl
bash tools/profile_bench.sh r03_u2netp --net u2netp --batch 12 --steps 20 --warmup 3 > gpurun_out/fb/profile_u2netp.txt 2>&1 || { tail gpurun_out/fb/profile_u2netp.txt; exit 1; }
head -12 gpurun_out/fb/profile_u2netp.txt
python - <<'PY'
import json
for l in open("gpurun_out/fb/other_nets.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:40], round(d["value"], 1), "Mmaps/s", round(d["ms_per_step"], 4), "ms", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
PY

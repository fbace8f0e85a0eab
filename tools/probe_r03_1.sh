set -e
mkdir -p gpurun_out/p1
python tools/microbench.py 56:16384 72 80 112 128 144 160 224:4096 256 288:2048 320:2048 > gpurun_out/p1/baseline.txt 2>&1
for mb in 32 64 96 128 192 256; do
  echo "== DCTS_SPLIT_CHUNK_MB=$mb" >> gpurun_out/p1/chunk_sweep.txt
  DCTS_SPLIT_CHUNK_MB=$mb python tools/microbench.py 288:2048:3 320:2048:3 >> gpurun_out/p1/chunk_sweep.txt 2>&1
done
python tools/mall_probe.py > gpurun_out/p1/mall.txt 2>&1

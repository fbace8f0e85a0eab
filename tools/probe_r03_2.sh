set -e
mkdir -p gpurun_out/p2
for spec in 72:9645 72:32768 80:7812 80:32768 112:3985 112:8192 128:3051 128:8192 144:2411 144:4992 144:8192 160:1953 160:4096; do
  python tools/microbench.py $spec 2>&1 | grep -v amdgpu
  python tools/g2_dev.py build_dev/libg2.so $spec 2>&1 | grep -v amdgpu
done > gpurun_out/p2/compare.txt 2>&1

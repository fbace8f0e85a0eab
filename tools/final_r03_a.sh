# round-3 evidence run A (GPU box): full GPU test-suite, default bench line, kernel trace of the bench, microbench of every shape
export TMPDIR=/tmp
mkdir -p gpurun_out/fa
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/fa/gputest.txt 2>&1 || { tail -30 gpurun_out/fa/gputest.txt; exit 1; }
tail -3 gpurun_out/fa/gputest.txt
python bench.py > gpurun_out/fa/bench_default.json 2> gpurun_out/fa/bench_default.err || { tail gpurun_out/fa/bench_default.err; exit 1; }
bash tools/profile_bench.sh r03_default > gpurun_out/fa/profile_bench.txt 2>&1 || { tail gpurun_out/fa/profile_bench.txt; exit 1; }
python tools/microbench.py 2 4 6 7 8 9 10 12 14 16 18 20 24 28 30 32 36 40 48 56 60 64 72 80 96 112 128 144 160 192 224:4096 256 288:2048 320:2048 56:16384 72:32768 144:8192 3 5 11 13 15 22 26 33 44 52 63 56x28 28x56 14x20 20x14 28x14 16x32 64x32 60x36 7x10 40x30 68 88 120 176 208 240 272 352 384 448 512 2>&1 | grep -v amdgpu > gpurun_out/fa/microbench.txt
tail -5 gpurun_out/fa/microbench.txt

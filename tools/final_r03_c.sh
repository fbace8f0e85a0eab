# round-3 evidence run C (GPU box): the default bench line once the PMC record carries this binary's hash (roofline.traffic non-null),
# and the full-configuration score / mask parity records (tests/mask_parity.py: identical activations scored by the GPU path and by the
# CPU oracle) for ResNet-50 and U2-Net-p at 288 and 320
export TMPDIR=/tmp
mkdir -p gpurun_out/fc
python bench.py > gpurun_out/fc/bench_default.json 2> gpurun_out/fc/bench_default.err || { tail gpurun_out/fc/bench_default.err; exit 1; }
python tests/mask_parity.py --net u2netp --batch_size 12 --limit 5 --input_size 288 --out gpurun_out/fc/mask_u2netp_288.txt > gpurun_out/fc/mask_u2netp_288.log 2>&1 || { tail gpurun_out/fc/mask_u2netp_288.log; exit 1; }
tail -3 gpurun_out/fc/mask_u2netp_288.txt
python tests/mask_parity.py --net u2netp --batch_size 12 --limit 5 --input_size 320 --out gpurun_out/fc/mask_u2netp_320.txt > gpurun_out/fc/mask_u2netp_320.log 2>&1 || { tail gpurun_out/fc/mask_u2netp_320.log; exit 1; }
tail -3 gpurun_out/fc/mask_u2netp_320.txt
python tests/mask_parity.py --net resnet_50 --batch_size 256 --limit 5 --out gpurun_out/fc/mask_resnet50.txt > gpurun_out/fc/mask_resnet50.log 2>&1 || { tail gpurun_out/fc/mask_resnet50.log; exit 1; }
tail -3 gpurun_out/fc/mask_resnet50.txt

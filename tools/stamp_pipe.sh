#!/bin/bash
# Diagnostic: s_memtime phase stamps of the pipelined fused kernel. Run on the GPU box:
# tools/stamp_pipe.sh [edge] [nmaps]   (expects build_dev/libstamps.so, built with
# -DDCTS_FUSED_STAMPS, or builds it)
set -e
edge=${1:-224}; nmaps=${2:-4096}
if [ ! -f build_dev/libstamps.so ]; then
  mkdir -p build_dev
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Wno-inline-asm -fno-slp-vectorize -DDCTS_DEV_FAST \
    -DDCTS_FUSED_STAMPS -o build_dev/libstamps.so dct_pruning_amd/csrc/dct_kernels.hip
fi
python3 - "$edge" "$nmaps" <<'PY'
import ctypes, sys, torch
edge, nmaps = int(sys.argv[1]), int(sys.argv[2])
lib = ctypes.CDLL("build_dev/libstamps.so")
i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
lib.dcts_energy_f32_ex.argtypes = [vp] + [i64] * 8 + [i32] * 3 + [vp, vp, ctypes.c_size_t, vp, i32]
x = torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda")); out = torch.empty(1, nmaps, device="cuda")
def run():
    rc = lib.dcts_energy_f32_ex(x.data_ptr(), 1, nmaps, edge, edge, *x.stride(), 0, nmaps, 0, out.data_ptr(), None, 0, None, 6)
    assert rc == 0, rc
run(); torch.cuda.synchronize()
lib.dcts_debug_fused_stamps(None, 1)
run(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib.dcts_debug_fused_stamps(buf, 0)
names = ["P2 barrier(pre-dump)", "P2 dump", "P2 barrier(post-dump)", "P2 butterflies", "P2 barrier(mid)",
         "P2 role transform+energy", "P1 wait vmcnt", "P1 barrier(top)", "P1 butterflies+DMA issue", "P1 barrier(mid)",
         "P1 role transform", "reduce", "glue"]
nroles = sum(1 for r in range(16) if any(buf[r * 16 + i] for i in range(16)))
tot = [sum(buf[r * 16 + i] for r in range(16)) for i in range(16)]
waves = nroles * min(nmaps, 256)
all_ = sum(tot)
maps_per_wg = nmaps / min(nmaps, 256)
print("pipe %dx%d, %d maps: per-wave average cycles per map" % (edge, edge, nmaps))
for n, t in zip(names, tot):
    print("  %-28s %10.0f  %5.1f%%" % (n, t / waves / maps_per_wg, 100.0 * t / all_))
print("  %-28s %10.0f" % ("total", all_ / waves / maps_per_wg))
for r in range(nroles):
    print("  role %2d: " % r + " ".join("%6.0f" % (buf[r * 16 + i] / min(nmaps, 256) / maps_per_wg) for i in range(13)))
PY

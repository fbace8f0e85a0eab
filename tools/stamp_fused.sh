#!/bin/bash
# Diagnostic: build libdctscore with s_memtime stamps in the fused kernel and print where a wave's
# cycles go per phase. Run on the GPU box: tools/stamp_fused.sh [edge] [nmaps]
set -e
edge=${1:-224}; nmaps=${2:-4096}
# the diagnostic library: build_dev/libdctscore_stamps.so if it was built in the container (build_dev/ travels
# with gpurun; the single-unit build takes minutes), otherwise built here
lib=build_dev/libdctscore_stamps.so
if [ ! -f "$lib" ]; then
  mkdir -p gpurun_out/stamps
  lib=gpurun_out/stamps/libdctscore_stamps.so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -Wno-inline-asm -fno-slp-vectorize -DDCTS_FUSED_STAMPS \
    -o "$lib" dct_pruning_amd/csrc/dct_kernels.hip dct_pruning_amd/csrc/tile2d.hip
fi
export DCTS_STAMPS_LIB="$lib"
python3 - "$edge" "$nmaps" <<'PY'
import ctypes, os, sys, torch
edge, nmaps = int(sys.argv[1]), int(sys.argv[2])
lib = ctypes.CDLL(os.environ["DCTS_STAMPS_LIB"])
i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
lib.dcts_energy_f32_ex.argtypes = [vp] + [i64] * 8 + [i32] * 3 + [vp, vp, ctypes.c_size_t, vp, i32]
x = torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda")); out = torch.empty(1, nmaps, device="cuda")
def run():
    rc = lib.dcts_energy_f32_ex(x.data_ptr(), 1, nmaps, edge, edge, *x.stride(), 0, nmaps, 0, out.data_ptr(), None, 0, None, 5)
    assert rc == 0, rc
run(); torch.cuda.synchronize()
lib.dcts_debug_fused_stamps(None, 1)
run(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib.dcts_debug_fused_stamps(buf, 0)
names = ["P1 wait vmcnt", "P1 barrier(top)", "P1 stage issue", "P1 butterflies", "P1 barrier(mid)", "P1 role transform",
         "P2 barrier(pre-dump)", "P2 dump", "P2 barrier(post-dump)", "P2 butterflies", "P2 barrier(mid)", "loop glue",
         "P2 role transform+energy", "reduce+store"]
nroles = sum(1 for r in range(16) if any(buf[r * 16 + i] for i in range(16)))
tot = [sum(buf[r * 16 + i] for r in range(16)) for i in range(16)]
waves = nroles * min(nmaps, 256)
all_ = sum(tot)
print("fused %dx%d, %d maps: per-wave average cycles per map (s_memtime ticks = shader cycles / wave count)" % (edge, edge, nmaps))
maps_per_wg = nmaps / min(nmaps, 256)
for n, t in zip(names, tot):
    print("  %-26s %10.0f  %5.1f%%" % (n, t / waves / maps_per_wg, 100.0 * t / all_))
print("  %-26s %10.0f" % ("total", all_ / waves / maps_per_wg))
PY

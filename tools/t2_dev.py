#!/usr/bin/env python3
"""Development driver for the tile2d kernel family (dct_pruning_amd/csrc/tile2d.hip built alone):
timing, Parseval check, optional phase stamps.

build (in the container, seconds):
  hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-inline-asm -Wno-pass-failed -fno-slp-vectorize \
        -DDCTS_T2_DEV [-DDCTS_T2_STAMPS] [-DDCTS_T2_BEARLY=k] -o build_dev/libt2[_stamps].so dct_pruning_amd/csrc/tile2d.hip
run (GPU box): tools/t2_dev.py build_dev/libt2.so [edge:nmaps ...]
"""
import ctypes
import sys

import torch

NAMES = ["A networks", "barrier #4", "finish+write set 0", "barrier #1", "transforms set 0", "barrier #2",
         "write set 1", "barrier #3", "issue early loads", "transforms set 1", "late loads+reduce"]


def main():
    lib = ctypes.CDLL(sys.argv[1])
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.t2_dev_run.argtypes = [vp, i64, ctypes.c_int, vp, vp]
    stamps = hasattr(lib, "t2_dev_stamps")
    specs = sys.argv[2:] or ["224:996", "224:4096"]
    for sp in specs:
        edge, nmaps = (int(v) for v in sp.split(":"))
        nbuf = max(1, min(8, int(600e6 // (nmaps * edge * edge * 4)) + 1))
        bufs = [torch.relu(torch.randn(nmaps, edge, edge, device="cuda")) for _ in range(nbuf)]
        out = torch.empty(nmaps, device="cuda")

        def run(x):
            rc = lib.t2_dev_run(x.data_ptr(), nmaps, edge, out.data_ptr(), None)
            assert rc == 0, rc

        for b in bufs:
            run(b)
        torch.cuda.synchronize()
        reps = 20
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for i in range(reps):
            ev[i][0].record()
            run(bufs[i % nbuf])
            ev[i][1].record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        med = ts[len(ts) // 2]
        x = bufs[(reps - 1) % nbuf]
        ref = (x.double() ** 2).sum(dim=(-2, -1))
        rel = ((out.double() - ref).abs() / ref.clamp_min(1e-30)).max().item()
        run(x)
        out2 = out.clone()
        run(x)
        same = bool((out2 == out).all())
        by = nmaps * (4 * edge * edge + 4)
        print("%4d maps=%-6d med %8.1f us min %8.1f us %8.1f GB/s %5.1f%% of 8 TB/s  relerr %.1e  bitwise-repeatable %s"
              % (edge, nmaps, med * 1e3, ts[0] * 1e3, by / med / 1e6, by / med / 1e6 / 80, rel, same), flush=True)
        if stamps:
            lib.t2_dev_stamps(None, 1)
            run(x)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 256)()
            lib.t2_dev_stamps(buf, 0)
            wgs = min(nmaps, 256)
            per_wg_maps = nmaps / wgs
            tot = [sum(buf[w * 16 + i] for w in range(16)) for i in range(16)]
            all_ = sum(tot[:11])
            print("  per-wave average cycles per map (16 waves, %d workgroups, %.1f maps each):" % (wgs, per_wg_maps))
            for n, t in zip(NAMES, tot):
                print("    %-22s %8.0f  %5.1f%%" % (n, t / 16 / wgs / per_wg_maps, 100.0 * t / all_))
            print("    %-22s %8.0f   (+ prologue %0.f per workgroup)" % ("total", all_ / 16 / wgs / per_wg_maps, tot[15] / 16 / wgs))
            for w in range(16):
                print("    wave %2d: " % w + " ".join("%6.0f" % (buf[w * 16 + i] / wgs / per_wg_maps) for i in range(11)))


if __name__ == "__main__":
    main()

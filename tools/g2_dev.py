#!/usr/bin/env python3
"""Development driver for the tile2g kernel family (dct_pruning_amd/csrc/tile2g.hip built alone): timing,
Parseval check, bit-repeatability, ragged map counts, optional coefficient check and phase stamps.

build (in the container; one shape compiles in about a minute):
  hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-inline-asm -Wno-pass-failed -fno-slp-vectorize \
        -DDCTS_G2_DEV [-DDCTS_G2_STAMPS] [-DDCTS_G2_NOSTORE] ['-DDCTS_TILE2G_TABLE(X)=X(144,3,18,3)'] \
        -o build_dev/libg2.so dct_pruning_amd/csrc/tile2g.hip
run (GPU box): tools/g2_dev.py build_dev/libg2.so [edge[:nmaps] ...] [--coeff]
"""
import ctypes
import sys

import torch

NAMES = ["A networks", "barrier top", "write set 0", "barrier", "passes set 0", "barrier", "write set 1", "barrier",
         "passes set 1", "-", "-", "-", "reduce", "load issue (both sets)"]


def main():
    lib = ctypes.CDLL(sys.argv[1])
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lib.g2_dev_run.argtypes = [vp, i64, ctypes.c_int, vp, vp]
    lib.g2_dev_coeff.argtypes = [vp, i64, ctypes.c_int, vp, vp, i64, vp]
    stamps = hasattr(lib, "g2_dev_stamps")
    args = [a for a in sys.argv[2:] if not a.startswith("--")]
    coeff = "--coeff" in sys.argv
    specs = args or ["144"]
    for sp in specs:
        parts = sp.split(":")
        edge = int(parts[0])
        nmaps = int(parts[1]) if len(parts) > 1 else max(64, int(200e6 // (edge * edge * 4)))
        nbuf = max(1, min(8, int(600e6 // (nmaps * edge * edge * 4)) + 1))
        bufs = [torch.relu(torch.randn(nmaps, edge, edge, device="cuda")) for _ in range(nbuf)]
        for b in bufs:
            b[5::8] = 0  # dead maps must come out as +0.0
        out = torch.full((nmaps,), -1.0, device="cuda")

        def run(x, n=nmaps, o=out):
            rc = lib.g2_dev_run(x.data_ptr(), n, edge, o.data_ptr(), None)
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
        rel = ((out.double() - ref).abs() / ref.clamp_min(1e-30))[ref > 0].max().item()
        dead_ok = bool((out[5::8] == 0).all() and not torch.signbit(out[5::8]).any())
        run(x)
        out2 = out.clone()
        run(x)
        same = bool((out2 == out).all())
        # ragged counts: 1, G-1, G+1 ... maps; the tail must not be written
        ragged_ok = True
        for n in (1, 2, 3, 5, 7, 257, 1025):
            if n > nmaps:
                continue
            o = torch.full((n + 3,), -7.0, device="cuda")
            run(x, n, o)
            torch.cuda.synchronize()
            r = ((o[:n].double() - ref[:n]).abs() / ref[:n].clamp_min(1e-30))[ref[:n] > 0]
            ragged_ok &= bool((o[n:] == -7.0).all()) and (r.numel() == 0 or r.max().item() < 1e-5) and bool((o[:n] == out[:n]).all())
        by = nmaps * (4 * edge * edge + 4)
        print("%4d maps=%-6d med %8.1f us min %8.1f us %8.1f GB/s %5.1f%% of 8 TB/s  relerr %.1e  dead+0 %s  repeatable %s  ragged %s"
              % (edge, nmaps, med * 1e3, ts[0] * 1e3, by / med / 1e6, by / med / 1e6 / 80, rel, dead_ok, same, ragged_ok), flush=True)
        if coeff:
            import numpy as np
            import scipy.fft
            n = 5
            xs = bufs[0][:n].contiguous()
            tile = edge * edge
            co = torch.full((n * tile + 4 * tile,), -3.0, device="cuda")        # guard words behind the output ...
            scratch = torch.full((4 * tile + 4 * tile,), -3.0, device="cuda")   # ... and behind the 4-tile scratch
            rc = lib.g2_dev_coeff(xs.data_ptr(), n, edge, co.data_ptr(), scratch.data_ptr(), 4, None)
            assert rc == 0, rc
            torch.cuda.synchronize()
            guards_ok = bool((co[n * tile:] == -3.0).all() and (scratch[4 * tile:] == -3.0).all())
            want = scipy.fft.dctn(xs.cpu().double().numpy(), type=2, norm="ortho", axes=(1, 2))
            err = np.abs(co[:n * tile].view(n, edge, edge).cpu().numpy() - want).max() / np.abs(want).max()
            print("      coefficients vs SciPy float64: max err / max coeff = %.2e   guard words intact %s" % (err, guards_ok), flush=True)
            assert guards_ok
        if stamps:
            lib.g2_dev_stamps(None, 1)
            run(x)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 256)()
            lib.g2_dev_stamps(buf, 0)
            tot = [sum(buf[w * 16 + i] for w in range(16)) for i in range(16)]
            all_ = sum(tot[:14])
            print("  per-wave average cycles per MAP (16 waves):")
            for nme, t in zip(NAMES, tot):
                if t:
                    print("    %-14s %8.0f  %5.1f%%" % (nme, t / 16 / nmaps, 100.0 * t / all_))
            print("    %-14s %8.0f" % ("total", all_ / 16 / nmaps))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-shape timing of dcts_energy_f32 (HIP events on the launch stream).
usage: tools/microbench.py [edge|HxW[:nmaps[:algo]] ...]   e.g. 56 224:4096 224:4096:1 56x28 28x56::9"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_pruning_amd as dpa  # noqa: E402

PEAK = 8000.0


def run(edge, nmaps, algo, reps=20, width=None):
    width = width or edge
    target = 600e6  # rotate buffers so the working set exceeds the 256 MiB Infinity Cache
    per = nmaps * edge * width * 4
    nbuf = max(1, min(8, int(target // per) + 1))
    bufs = [torch.relu(torch.randn(1, nmaps, edge, width, device="cuda")) for _ in range(nbuf)]
    out = torch.empty(1, nmaps, device="cuda")
    for b in bufs:
        dpa.energy_nc(b, algo=algo, out=out)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        dpa.energy_nc(bufs[i % nbuf], algo=algo, out=out)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    med = ts[len(ts) // 2]
    by = nmaps * (4 * edge * width + 4)
    ref = (bufs[(reps - 1) % nbuf].double() ** 2).sum(dim=(-2, -1))
    rel = ((out.double() - ref).abs() / ref.clamp_min(1e-30)).max().item()
    print("%4dx%-4d maps=%-8d algo=%d  med %9.1f us  min %9.1f us  %9.2f Mmaps/s  %7.1f GB/s  %5.1f%% of 8TB/s  relerr %.1e"
          % (edge, width, nmaps, algo, med * 1e3, ts[0] * 1e3, nmaps / med / 1e3, by / med / 1e6, by / med / 1e6 / PEAK * 100, rel),
          flush=True)


if __name__ == "__main__":
    specs = sys.argv[1:] or ["56", "224"]
    for sp in specs:
        parts = sp.split(":")
        hw = parts[0].split("x")
        edge, width = int(hw[0]), int(hw[-1])
        nmaps = int(parts[1]) if len(parts) > 1 and parts[1] else max(64, int(200e6 // (edge * width * 4)))
        algo = int(parts[2]) if len(parts) > 2 else 0
        run(edge, nmaps, algo, width=width)

#!/usr/bin/env python3
"""Timing + parity of a development build: tools/devbench.py <lib.so> edge:nmaps:algo ...
(DEVBENCH_REPS=100: the headline protocol - 20 warm-up launches, then that many timed ones)"""
import ctypes
import os
import sys

import torch

lib = ctypes.CDLL(sys.argv[1])
i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
lib.dcts_energy_f32_ex.argtypes = [vp] + [i64] * 8 + [i32] * 3 + [vp, vp, ctypes.c_size_t, vp, i32]
lib.dcts_workspace_bytes.restype = ctypes.c_size_t
lib.dcts_workspace_bytes.argtypes = [i64] * 4
for spec in sys.argv[2:]:
    edge, nmaps, algo = (int(v) for v in spec.split(":"))
    nbuf = max(1, min(8, int(600e6 // (nmaps * edge * edge * 4)) + 1))
    bufs = [torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda")) for _ in range(nbuf)]
    out = torch.empty(1, nmaps, device="cuda")
    need = lib.dcts_workspace_bytes(1, nmaps, edge, edge)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")

    def run(x):
        rc = lib.dcts_energy_f32_ex(x.data_ptr(), 1, nmaps, edge, edge, *x.stride(), 0, nmaps, 0, out.data_ptr(),
                                    ws.data_ptr(), ws.numel(), None, algo)
        assert rc == 0, rc

    for b in bufs:
        run(b)
    torch.cuda.synchronize()
    reps = int(os.environ.get("DEVBENCH_REPS", "20"))
    if reps > 20:
        for i in range(20):
            run(bufs[i % nbuf])
        torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        run(bufs[i % nbuf])
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    med = ts[len(ts) // 2]
    ref = (bufs[(reps - 1) % nbuf].double() ** 2).sum(dim=(-2, -1))
    rel = ((out.double() - ref).abs() / ref.clamp_min(1e-30)).max().item()
    by = nmaps * (4 * edge * edge + 4)
    print("%4d maps=%-7d algo=%d med %8.1f us min %8.1f us %8.1f GB/s %5.1f%% relerr %.1e" %
          (edge, nmaps, algo, med * 1e3, ts[0] * 1e3, by / med / 1e6, by / med / 1e6 / 80, rel), flush=True)

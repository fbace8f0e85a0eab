#!/usr/bin/env python3
"""Per-repetition launch times of dcts_energy_f32 in issue order (HIP events), to look at run-to-run
structure (clock ramps, bimodal launches). usage: tools/rep_times.py edge:nmaps[:algo[:reps[:nbuf]]] ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_pruning_amd as dpa  # noqa: E402

for sp in sys.argv[1:]:
    parts = [int(v) for v in sp.split(":")]
    edge, nmaps = parts[0], parts[1]
    algo = parts[2] if len(parts) > 2 else 0
    reps = parts[3] if len(parts) > 3 else 40
    nbuf = parts[4] if len(parts) > 4 else 1
    bufs = [torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda")) for _ in range(nbuf)]
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
    ts = [a.elapsed_time(b) * 1e3 for a, b in ev]
    print("%dx%d maps=%d algo=%d nbuf=%d: " % (edge, edge, nmaps, algo, nbuf) + " ".join("%.0f" % t for t in ts), flush=True)

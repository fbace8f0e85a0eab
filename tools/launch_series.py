#!/usr/bin/env python3
"""A back-to-back series of launches of one shape as bench.py's headline runs them (20 warm-up + 100 timed, three
rotating buffers), for `rocprofv3 --kernel-trace`: prints the HIP-event time of every launch; tools/series_gaps.py
then sets the profiler's per-dispatch durations and the gaps between dispatches beside them (where do the 4-6 us per
launch of the 56 x 56 headline go?).  usage: tools/launch_series.py edge maps [reps] [events 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_pruning_amd as dpa  # noqa: E402

edge, nmaps = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
with_events = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
nbuf = 3
bufs = [torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda")) for _ in range(nbuf)]
out = torch.empty(1, nmaps, device="cuda")
for i in range(20):
    dpa.energy_nc(bufs[i % nbuf], out=out)
torch.cuda.synchronize()
if with_events:
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        ev[i][0].record()
        dpa.energy_nc(bufs[i % nbuf], out=out)
        ev[i][1].record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) * 1e3 for a, b in ev]
    print("event_us " + " ".join("%.1f" % t for t in ts))
    s = sorted(ts)
    by = nmaps * (4 * edge * edge + 4)
    print("events: median %.1f us  min %.1f us  -> %.1f %% / %.1f %% of 8 TB/s" % (s[len(s) // 2], s[0], by / s[len(s) // 2] / 80e3, by / s[0] / 80e3))
else:
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for i in range(reps):
        dpa.energy_nc(bufs[i % nbuf], out=out)
    t1.record()
    torch.cuda.synchronize()
    per = t0.elapsed_time(t1) * 1e3 / reps
    print("no per-launch events: %.1f us per launch over %d launches -> %.1f %% of 8 TB/s" % (per, reps, nmaps * (4 * edge * edge + 4) / per / 80e3))

#!/usr/bin/env python3
"""Soak for races: the 2-D split, pipelined, fused and lane-per-map kernels and the wide-grid codelet kernels
on random map counts, twice each (must be bit-identical) and against sum(x^2) (Parseval, 1e-5).
usage: tools/soak.py [iterations]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_pruning_amd as dpa  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(7)
bad = 0
for it in range(iters):
    edge, algo, hi = [(224, dpa.ALGO_PIPE, 1500), (128, dpa.ALGO_PIPE, 4000), (7, dpa.ALGO_LANE, 300000),
                      (9, dpa.ALGO_LANE, 200000), (224, dpa.ALGO_FUSED, 1500), (256, dpa.ALGO_FUSED, 1200), (288, dpa.ALGO_FUSED, 900),
                      (320, dpa.ALGO_FUSED, 700), (224, dpa.ALGO_TILE2D, 3000), (224, dpa.ALGO_TILE2D, 700),
                      (56, dpa.ALGO_AUTO, 60000), (28, dpa.ALGO_AUTO, 200000), (14, dpa.ALGO_AUTO, 600000),
                      # round 3: several maps per round (tile2g.hip), incl. the two-workgroups-per-CU shapes and the odd pad
                      (72, dpa.ALGO_TILE2D, 20000), (80, dpa.ALGO_TILE2D, 15000), (144, dpa.ALGO_TILE2D, 6000),
                      (160, dpa.ALGO_TILE2D, 4000), (128, dpa.ALGO_TILE2D, 6000), (112, dpa.ALGO_TILE2D, 8000),
                      (71, dpa.ALGO_AUTO, 20000), (143, dpa.ALGO_AUTO, 6000), (60, dpa.ALGO_AUTO, 40000), (48, dpa.ALGO_AUTO, 60000),
                      # the run-time codelet pair (rect.hip): odd / prime edges, a tabulated edge through it; a two-launch edge of round 3
                      (13, dpa.ALGO_AUTO, 300000), (22, dpa.ALGO_AUTO, 100000), (63, dpa.ALGO_AUTO, 12000), (28, dpa.ALGO_RECT, 60000),
                      (176, dpa.ALGO_AUTO, 1500)][it % 28]
    nmaps = int(rng.integers(1, hi))
    pad = edge % 2 == 1 and edge > 9
    x = torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda"))
    a = dpa.energy_nc(x, algo=algo, pad_front_if_odd=pad)
    b = dpa.energy_nc(x, algo=algo, pad_front_if_odd=pad)
    ref = (x.double() ** 2).sum(dim=(-2, -1))
    rel = ((a.double() - ref).abs() / ref.clamp_min(1e-30)).max().item()
    if not torch.equal(a, b) or not rel <= 1e-5:
        bad += 1
        print("MISMATCH edge %d algo %d nmaps %d: equal=%s rel=%g" % (edge, algo, nmaps, torch.equal(a, b), rel), flush=True)
    if it % 50 == 0:
        print("iter", it, "ok so far, bad =", bad, flush=True)
print("soak done: %d iterations, %d bad" % (iters, bad))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""Run one shape/algo a few times (for PMC passes). usage: probe_one.py edge nmaps algo"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_pruning_amd as dpa  # noqa: E402

edge, nmaps, algo = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
t = torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda"))
for _ in range(3):
    dpa.energy_nc(t, algo=algo)
torch.cuda.synchronize()
print("done")

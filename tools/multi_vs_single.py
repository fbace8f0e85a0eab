#!/usr/bin/env python3
"""Kernel time of a multi-tensor launch (U2-Net-p's tensors of one tile shape at batch 12) against one tensor with the same number of
maps: run under `rocprofv3 --kernel-trace --stats` and read the per-kernel averages (the Python side of energy_multi is host-bound)."""
import sys
import torch
sys.path.insert(0, '.')
import dct_pruning_amd as dpa
for edge, chans in [(72, [64, 16, 16, 64, 16, 16, 64, 64, 16, 16, 64, 16, 16, 16, 16, 16]), (144, [64, 16, 16, 64, 16, 64, 16, 16, 64, 16, 16, 16, 16]),
                    (288, [64, 16, 64, 64, 16, 64, 64]), (224, [16] * 20)]:
    tens = [torch.relu(torch.randn(12, c, edge, edge, device='cuda')) for c in chans]
    total = sum(12 * c for c in chans)
    one = torch.relu(torch.randn(1, total, edge, edge, device='cuda'))
    items = [(t, 0, None) for t in tens]
    for _ in range(20):
        dpa.energy_multi(items)
    torch.cuda.synchronize()
    for _ in range(20):
        dpa.energy_nc(one)
    torch.cuda.synchronize()
    print(edge, 'maps', total, 'tensors', len(chans))

#!/usr/bin/env python3
"""Workload for the PMC passes (run under rocprofv3 --pmc ...): a calibration read of known
size in the kernels' access width, a torch copy (16 B/lane), then the headline shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_pruning_amd as dpa  # noqa: E402
from dct_pruning_amd import _lib  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
n = 1 << 28  # 1 GiB of floats: far beyond L2 + Infinity Cache
x = torch.ones(n, device=dev)
sink = torch.zeros(4, device=dev)
for _ in range(3):
    _lib.check(lib.dcts_debug_stream_read_f32(x.data_ptr(), n, sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
del y
torch.cuda.synchronize()
for edge, nmaps in [(56, 65536), (28, 262144), (14, 1048576), (7, 4194304), (32, 196608), (224, 4096), (128, 12288),
                    (288, 2048)]:
    t = torch.relu(torch.randn(1, nmaps, edge, edge, device=dev))
    for _ in range(3):
        dpa.energy_nc(t)
    if edge in (224, 128):  # AUTO runs the pipelined kernel there: profile the plain fused one as well
        for _ in range(3):
            dpa.energy_nc(t, algo=dpa.ALGO_FUSED)
    torch.cuda.synchronize()
    del t
print("probe done")

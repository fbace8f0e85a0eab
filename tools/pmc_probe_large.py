#!/usr/bin/env python3
"""Workload for the PMC passes of the mid-size / large-tile kernels (run under rocprofv3 --pmc ..., tools/profile_pmc_large.sh):
a calibration read of known size (one dword per lane, the gather kernels' width; the guide gives the same factor for the
16-byte direct-to-LDS loads of the fused kernels), then three AUTO launches per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dct_pruning_amd as dpa  # noqa: E402
from dct_pruning_amd import _lib  # noqa: E402

SHAPES = [(72, 32768), (80, 24576), (96, 16384), (112, 12288), (128, 12288), (144, 8192), (160, 6144), (192, 4096), (224, 4096),
          (256, 3072), (288, 2048), (320, 2048)]

if __name__ == "__main__":
    lib = _lib.load()
    dev = torch.device("cuda:0")
    n = 1 << 28  # 1 GiB of floats: far beyond L2 + Infinity Cache
    x = torch.ones(n, device=dev)
    sink = torch.zeros(4, device=dev)
    for _ in range(3):
        _lib.check(lib.dcts_debug_stream_read_f32(x.data_ptr(), n, sink.data_ptr(), torch.cuda.current_stream().cuda_stream))
    del x
    torch.cuda.synchronize()
    for edge, nmaps in SHAPES:
        t = torch.relu(torch.randn(1, nmaps, edge, edge, device=dev))
        for _ in range(3):
            dpa.energy_nc(t)
        torch.cuda.synchronize()
        del t
    print("probe done")

#!/usr/bin/env python3
"""Two development builds on the same inputs: are the energies bit-identical? tools/devcmp.py <a.so> <b.so> edge:nmaps ..."""
import ctypes
import sys

import torch


def load(path):
    lib = ctypes.CDLL(path)
    i64, i32, vp = ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p
    lib.dcts_energy_f32_ex.argtypes = [vp] + [i64] * 8 + [i32] * 3 + [vp, vp, ctypes.c_size_t, vp, i32]
    lib.dcts_workspace_bytes.restype = ctypes.c_size_t
    lib.dcts_workspace_bytes.argtypes = [i64] * 4
    return lib


a, b = load(sys.argv[1]), load(sys.argv[2])
for spec in sys.argv[3:]:
    edge, nmaps = (int(v) for v in spec.split(":"))
    x = torch.relu(torch.randn(1, nmaps, edge, edge, device="cuda"))
    outs = []
    for lib in (a, b):
        out = torch.empty(1, nmaps, device="cuda")
        need = lib.dcts_workspace_bytes(1, nmaps, edge, edge)
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda")
        rc = lib.dcts_energy_f32_ex(x.data_ptr(), 1, nmaps, edge, edge, *x.stride(), 0, nmaps, 0, out.data_ptr(), ws.data_ptr(),
                                    ws.numel(), None, 0)
        assert rc == 0, rc
        torch.cuda.synchronize()
        outs.append(out)
    print("%d x %d, %d maps: bit-identical %s" % (edge, edge, nmaps, torch.equal(outs[0], outs[1])))

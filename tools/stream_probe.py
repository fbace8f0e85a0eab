import os, sys, torch
sys.path.insert(0, "/root/repo")
from dct_pruning_amd import _lib
lib = _lib.load()
n = 1 << 28
x = torch.ones(n, device="cuda"); sink = torch.zeros(4, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for sz in [1<<26, 1<<28]:
    for _ in range(3): lib.dcts_debug_stream_read_f32(x.data_ptr(), sz, sink.data_ptr(), st)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); lib.dcts_debug_stream_read_f32(x.data_ptr(), sz, sink.data_ptr(), st); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print("dword stream read %d MB: med %.1f us -> %.0f GB/s" % (sz*4>>20, ts[5]*1e3, sz*4/ts[5]/1e6))
y = torch.empty(1<<27, device="cuda")
for _ in range(3): y.copy_(x[:1<<27])
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); y.copy_(x[:1<<27]); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print("torch copy 512MB->512MB: med %.1f us -> %.0f GB/s (r+w)" % (ts[5]*1e3, 2*(1<<29)/ts[5]/1e6))
s = x[:1<<27]
for _ in range(3): s.sum()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
for a, b in ev:
    a.record(); s.sum(); b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print("torch sum 512MB: med %.1f us -> %.0f GB/s" % (ts[5]*1e3, (1<<29)/ts[5]/1e6))

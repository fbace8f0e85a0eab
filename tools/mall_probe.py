import torch, time
def bench(nbytes, reps=20):
    n = nbytes // 4
    x = torch.ones(n, device="cuda"); y = torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); y.copy_(x); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    print("copy %4d MB -> %4d MB: med %.1f us  %.0f GB/s (r+w)" % (nbytes>>20, nbytes>>20, ts[len(ts)//2]*1e3, 2*nbytes/ts[len(ts)//2]/1e6))
    # write then read same buffer alternately: y.fill_ then y.sum
    for _ in range(3): y.fill_(1.0); y.sum()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b, c in ev:
        a.record(); y.fill_(2.0); b.record(); y.sum(); c.record()
    torch.cuda.synchronize()
    tw = sorted(a.elapsed_time(b) for a, b, c in ev); tr = sorted(b.elapsed_time(c) for a, b, c in ev)
    print("   fill %.1f us (%.0f GB/s)   sum-after-fill %.1f us (%.0f GB/s)" % (tw[len(tw)//2]*1e3, nbytes/tw[len(tw)//2]/1e6, tr[len(tr)//2]*1e3, nbytes/tr[len(tr)//2]/1e6))
for mb in (16, 32, 64, 100, 200, 512, 1024):
    bench(mb << 20)

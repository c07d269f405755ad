"""Practical HBM read bandwidth of the GPU it runs on (torch reductions over a buffer far larger than the caches), to put next
to the 8 TB/s datasheet peak the rooflines are priced against: python tools/hbm_read_probe.py [GiB]"""
import sys
import torch

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
n = int(gib * 2 ** 30) // 4
x = torch.ones(n, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for name, fn, nbytes in (("sum  (read only)", lambda: x.sum(), 4 * n), ("copy (read + write)", lambda: y.copy_(x), 8 * n),
                         ("fill (write only)", lambda: y.fill_(1.0), 4 * n)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("%-20s %.3f ms  %.2f TB/s" % (name, best, nbytes / best / 1e9))

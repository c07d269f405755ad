import time, torch
torch.cuda.init()
n = 2 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("pinned H2D %.1f GB/s" % (n / dt / 1e9))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda"); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
torch.cuda.synchronize(); t0 = time.perf_counter()
with torch.cuda.stream(s1): d.copy_(h, non_blocking=True)
with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("two streams H2D %.1f GB/s" % (2 * n / dt / 1e9))
import numpy as np, threading
src = np.ones(n, np.uint8); dst = h.numpy()
def cp(a, b):
    dst[a:b] = src[a:b]
for nt in (8, 32, 64):
    t0 = time.perf_counter()
    th = [threading.Thread(target=cp, args=(i * n // nt, (i + 1) * n // nt)) for i in range(nt)]
    [t.start() for t in th]; [t.join() for t in th]
    print("host memcpy into pinned, %d threads: %.1f GB/s" % (nt, n / (time.perf_counter() - t0) / 1e9))

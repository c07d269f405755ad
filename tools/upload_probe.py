import sys, time, os
sys.path.insert(0, "/root/repo/pixel-perfect-sfm_amd")
import numpy as np, torch
from pixsfm_amd.engine import Context, PatchArena
ctx = Context(0)
n = 131072                                   # 8.6 GB of 16x16x128 fp16 patches
host = np.empty((n, 16, 16, 128), np.float16); host[:] = 1.0
corners = np.zeros((n, 2), np.int32); scales = np.ones((n, 2))
a = PatchArena(ctx, n, 16, 16, 128, np.float16)
for rep in range(3):
    t0 = time.perf_counter(); a.upload(0, host, corners, scales); ctx.sync(); dt = time.perf_counter() - t0
    print("contiguous upload %.1f GB/s" % (host.nbytes / dt / 1e9))
ptrs = host.ctypes.data + np.arange(n, dtype=np.uint64) * np.uint64(host[0].nbytes)
perm = np.random.default_rng(0).permutation(n)
for name, pp in (("in order", ptrs), ("permuted", ptrs[perm])):
    t0 = time.perf_counter(); b = PatchArena.from_patch_pointers(ctx, pp, (16, 16, 128), np.float16, corners, scales); ctx.sync(); dt = time.perf_counter() - t0
    print("gather upload, %s: %.1f GB/s" % (name, host.nbytes / dt / 1e9)); del b

"""GPU: time pxr_dense_spd_solve at the reduced-system size of the bench problem (run under rocprofv3 --kernel-trace --stats for per-kernel numbers)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pixel-perfect-sfm_amd"))
from pixsfm_amd.engine import Context   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1593
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = Context(0)
rng = np.random.default_rng(0)
M = rng.normal(size=(n, n + 5))
A = np.triu(M @ M.T + np.eye(n) * 1e-3 * n)
b = rng.normal(size=n)
info = C.c_int(0)
dA, db = ctx.to_device(A), ctx.to_device(b)
for _ in range(3):
    dA.upload(A); db.upload(b)
    ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info))
ctx.sync()
tot = 0.0
for _ in range(reps):
    dA.upload(A); db.upload(b)
    ctx.sync()
    t0 = time.perf_counter()
    ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info))
    ctx.sync()
    tot += time.perf_counter() - t0
print("n = %d: %.3f ms per factor + solve (incl. pack / unpack), info %d" % (n, 1e3 * tot / reps, info.value))

import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pixel-perfect-sfm_amd"))
import numpy as np, ctypes as C
from pixsfm_amd.engine import Context
from pixsfm_amd._lib import check
ctx = Context(0)
rng = np.random.default_rng(0)
for n in (130, 1593, 6000):
    M = rng.normal(size=(n, n)); A = M @ M.T + n * np.eye(n); b = rng.normal(size=n)
    dA = ctx.to_device(A, np.float64); db = ctx.to_device(b, np.float64)
    info = C.c_int()
    for rep in range(3):
        dA.upload(A); db.upload(b)
        ctx.sync(); t0 = time.perf_counter()
        check(ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info)), "solve")
        dt = time.perf_counter() - t0
    x = db.download()
    print(n, info.value, "resid", np.abs(A @ x - b).max() / np.abs(b).max(), "wall_ms", dt * 1e3, "GF/s", n**3 / 3 / dt / 1e9)

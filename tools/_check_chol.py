"""GPU: pxr_dense_spd_solve against numpy at a few sizes (python tools/_check_chol.py 1593 2500 5000)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pixel-perfect-sfm_amd"))
from pixsfm_amd.engine import Context   # noqa: E402

ctx = Context(0)
for n in [int(a) for a in sys.argv[1:]] or [1593]:
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n + 5))
    A = M @ M.T + np.eye(n) * 1e-3 * n
    b = rng.normal(size=n)
    info = C.c_int(0)
    dA, db = ctx.to_device(np.triu(A)), ctx.to_device(b)
    rc = ctx.lib.pxr_dense_spd_solve(ctx.handle, dA.ptr, n, db.ptr, C.byref(info))
    x = db.download()
    want = np.linalg.solve(A, b)
    Lf = np.triu(dA.download()).T
    Lw = np.linalg.cholesky(A)
    print("n = %5d rc %d info %d  |x - x*| / |x*| = %.3e   |L - L*| / |L*| = %.3e" % (n, rc, info.value, np.linalg.norm(x - want) / np.linalg.norm(want),
                                                                                 np.linalg.norm(Lf - Lw) / np.linalg.norm(Lw)))

"""Generates tests/golden/irls_ref.npz by running the REFERENCE's own RobustMeanIRLS (pixsfm/base/src/irls_optim.h,
compiled in place into oracle/_ref/libpxo_ref_irls.so by oracle/Makefile against a minimal matrix class -- Eigen is
absent here) on seeded descriptor tracks.

Run in the build container only (needs /root/reference):  make -C oracle && python tests/golden/make_golden_irls.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libpxo_ref_irls.so")
LOSSES = {"trivial": 0, "cauchy": 1, "huber": 2}


def cases():
    """(name, descs (n, C), loss name, scale, iters, l2_normalize) -- seeded."""
    rng = np.random.default_rng(271828)
    out = []
    k = 0
    for n in (1, 2, 3, 5, 8, 20):
        for ch in (128, 64, 3):
            base = rng.normal(0, 1, ch); base /= np.linalg.norm(base)
            d = base + rng.normal(0, 0.05, (n, ch))
            n_out = n // 4
            if n_out:
                d[rng.choice(n, n_out, replace=False)] = rng.normal(0, 1, (n_out, ch))     # outliers
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            loss, a = [("cauchy", 0.25), ("cauchy", 0.25), ("huber", 0.3), ("trivial", 1.0)][k % 4]
            iters = [100, 100, 7, 3][k % 4]
            out.append(("irls%02d" % k, d, loss, a, iters, k % 5 != 4))
            k += 1
    same = np.tile(out[0][1][:1], (4, 1))
    out.append(("identical", same, "cauchy", 0.25, 100, True))        # rho = 0: the early return of irls_optim.h:60-67
    return out


def run_reference(d, loss, a, iters, l2):
    lib = C.CDLL(LIB)
    d = np.ascontiguousarray(d, dtype=np.float64)
    mean = np.empty(d.shape[1])
    early = lib.pxo_ref_robust_mean_irls(C.c_void_p(d.ctypes.data), d.shape[0], d.shape[1], LOSSES[loss], C.c_double(a),
                                         int(iters), int(bool(l2)), C.c_void_p(mean.ctypes.data))
    return mean, early


if __name__ == "__main__":
    store = {}
    for name, d, loss, a, iters, l2 in cases():
        mean, early = run_reference(d, loss, a, iters, l2)
        store[name + "_mean"] = mean
        store[name + "_early"] = np.array([early])
    path = os.path.join(HERE, "irls_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, len(cases()), "tracks")

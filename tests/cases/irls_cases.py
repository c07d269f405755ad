"""Seeded descriptor tracks for the robust mean (base/src/irls_optim.h:24-71).  Inputs only."""
import numpy as np


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

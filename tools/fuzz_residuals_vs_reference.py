"""Randomised CPU sweep: the ORACLE's residual functors (KA edge, unary reference, BA residual with every Jacobian block)
against the REFERENCE's own functors compiled in place (oracle/_ref/libpxo_ref_residual.so), on the case generators of
tests/golden/make_golden_residuals.py re-seeded.  Build container only; not part of the test suite.
python tools/fuzz_residuals_vs_reference.py [n_seeds] [first_seed]"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
spec = importlib.util.spec_from_file_location("g", os.path.join(ROOT, "tests", "golden", "make_golden_residuals.py"))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)
import pxo  # noqa: E402


def rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
worst = {"ka": 0.0, "unary": 0.0, "ba_r": 0.0, "ba_J": 0.0}
n = 0
for seed in range(first, first + n_seeds):
    for c in G.ka_cases(seed):
        r, J1, J2, r2, Jk = G.run_ka(c)
        cfg = pxo.cfg(c["l2"], c["float_simd"], False)
        p1, p2 = pxo.make_patch(c["d1"], c["c1"], c["s1"]), pxo.make_patch(c["d2"], c["c2"], c["s2"])
        ro, J1o, J2o = pxo.ka_residual(p1, p2, cfg, c["kp1"], c["kp2"])
        worst["ka"] = max(worst["ka"], rel(ro, r), rel(J1o, J1), rel(J2o, J2))
        r2o, Jko = pxo.ref2d_residual(p1, cfg, c["kp1"], c["ref"])
        worst["unary"] = max(worst["unary"], rel(r2o, r2), rel(Jko, Jk))
        n += 1
    for c in G.ba_cases(seed + 100000):
        r, J, ok = G.run_ba(c, False)
        patch = pxo.make_patch(c["d"], c["c"], c["s"])
        ro, Jq, Jt, JX, Jk = pxo.ba_residual(patch, pxo.cfg(c["l2"], False, c["check_bounds"]), c["model"], c["q"], c["t"], c["X"], c["params"], c["ref"])
        worst["ba_r"] = max(worst["ba_r"], rel(ro, r))
        worst["ba_J"] = max(worst["ba_J"], rel(np.hstack([Jq, Jt, JX, Jk]), J))
        n += 1
print("cases %d  worst relative differences %s" % (n, {k: float("%.2e" % v) for k, v in worst.items()}))

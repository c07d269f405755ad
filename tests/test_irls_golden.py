"""Robust-mean IRLS (A19) pinned against the REFERENCE's own loop: tests/golden/irls_ref.npz holds what
pixsfm/base/src/irls_optim.h:24-71 (compiled in place, tests/golden/make_golden_irls.py) returns for seeded
descriptor tracks -- weight normalisation, row normalisation of the mean, weights = 1 / rho(|d - mean|^2) from the
loss VALUE, the early return on rho <= 0.  The oracle's C restatement (oracle/pxo_geom.c) is checked against it here;
the GPU kernel is checked against the oracle in tests/test_refs_gpu.py.  (The in-place build uses a minimal matrix class
instead of Eigen, so sums run left to right: agreement to rounding, not bit-exact; the rho formulas are [upstream Ceres].)"""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_irls", os.path.join(HERE, "golden", "make_golden_irls.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_irls_matches_the_reference_vectors():
    import pxo
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "irls_ref.npz"))
    n_early = 0
    for name, d, loss, a, iters, l2 in gen.cases():
        mean, early = pxo.robust_mean_irls(d, pxo.loss(loss, a), iters=iters, l2_normalize=l2)
        want, want_early = gold[name + "_mean"], int(gold[name + "_early"][0])
        assert np.abs(mean - want).max() < 1e-13, (name, np.abs(mean - want).max())
        assert int(early >= 0) == want_early, name                        # the oracle returns the index of the early-return observation, -1 otherwise
        n_early += want_early
        # the reference of the point = the observation closest to that mean (reference_extractor.h:249-272)
        idx, ref, _ = pxo.compute_reference(d, pxo.loss(loss, a), iters=iters, l2_normalize=l2)
        assert idx == int(np.argmin(((d - want) ** 2).sum(1))) and np.array_equal(ref, d[idx])
    assert n_early >= 1


def test_reference_run_live_when_present():
    import pxo
    gen = _gen()
    if not os.path.exists(gen.LIB):
        pytest.skip("oracle/_ref/libpxo_ref_irls.so not built (reference tree absent)")
    rng = np.random.default_rng(4)
    for _ in range(30):
        n, ch = int(rng.integers(1, 12)), int(rng.choice([3, 16, 128]))
        d = rng.normal(0, 1, (n, ch)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        a = float(rng.choice([0.1, 0.25, 1.0]))
        want, _ = gen.run_reference(d, "cauchy", a, 100, True)
        got, _ = pxo.robust_mean_irls(d, pxo.loss("cauchy", a), iters=100, l2_normalize=True)
        assert np.abs(got - want).max() < 1e-12

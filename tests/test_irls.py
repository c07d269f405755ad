"""Robust-mean IRLS (pixsfm/base/src/irls_optim.h:24-71): the oracle's C restatement (oracle/pxo_geom.c)
against what the loop DEFINES, checked with numpy and independent of how the loop is written:
  * one step of the recurrence applied to the returned mean reproduces it (fixed point) once the iteration has converged:
        w_i = 1 / rho(|d_i - mu|^2)[0],  w <- w / sum w,  mu' = sum w_i d_i,  mu' <- mu' / |mu'|   (l2_normalize)
    -- with the loss VALUE rho[0], not rho' (irls_optim.h:60-67);
  * a single observation is its own mean; identical observations take the early return on rho = 0 (irls_optim.h:60-67);
  * with few iterations the result equals a direct numpy transcription of the recurrence run for that many steps.
The reference's own test (irls_optim_test.cc) only compares its static and dynamic template instantiations with each other
and holds no expected values; the header cannot be compiled here (Eigen / Ceres absent): PARITY UNPINNED.
The GPU kernel is checked against the oracle in tests/test_refs_gpu.py."""
import numpy as np

from cases import irls_cases as gen


def _rho(loss, a, s):
    """Published Ceres loss VALUES rho(s) (loss_function.h): trivial s; Cauchy b log(1 + s / b), b = a^2;
    Huber s <= b ? s : 2 a sqrt(s) - b."""
    if loss == "trivial":
        return s
    b = a * a
    if loss == "cauchy":
        return b * np.log1p(s / b)
    return np.where(s <= b, s, 2.0 * a * np.sqrt(s) - b)


def _step(d, mu, loss, a, l2):
    w = 1.0 / _rho(loss, a, ((d - mu) ** 2).sum(1))
    w = w / w.sum()
    m = (w[:, None] * d).sum(0)
    return m / np.linalg.norm(m) if l2 else m


def _numpy_irls(d, loss, a, iters, l2):
    w = np.ones(len(d))
    mu = None
    for _ in range(iters):
        w = w / w.sum()
        mu = (w[:, None] * d).sum(0)
        if l2:
            mu = mu / np.linalg.norm(mu)
        rho = _rho(loss, a, ((d - mu) ** 2).sum(1))
        if (rho <= 0).any():
            return d[int(np.argmax(rho <= 0))], True
        w = 1.0 / rho
    return mu, False


def test_oracle_irls_is_a_fixed_point_of_its_recurrence_and_equals_a_numpy_transcription():
    import pxo
    n_early = n_fixed = 0
    for name, d, loss, a, iters, l2 in gen.cases():
        mean, early = pxo.robust_mean_irls(d, pxo.loss(loss, a), iters=iters, l2_normalize=l2)
        want, want_early = _numpy_irls(d, loss, a, iters, l2)
        collapsed = np.abs(d - mean).max(1).min() < 1e-12        # the mean sits ON an observation: rho reaches 0 or 1e-30, by rounding
        assert (early >= 0) == want_early or collapsed, name
        assert np.abs(mean - want).max() < 1e-12, (name, np.abs(mean - want).max())
        n_early += int(want_early)
        if iters == 100 and not want_early and len(d) > 1 and not collapsed:
            assert np.abs(_step(d, mean, loss, a, l2) - mean).max() < 1e-9, name       # converged: one more step changes nothing
            n_fixed += 1
        if len(d) == 1:
            assert np.abs(mean - (d[0] / np.linalg.norm(d[0]) if l2 else d[0])).max() < 1e-15
        # the reference of the point = the observation closest to that mean (reference_extractor.h:249-272)
        idx, ref, _ = pxo.compute_reference(d, pxo.loss(loss, a), iters=iters, l2_normalize=l2)
        assert idx == int(np.argmin(((d - mean) ** 2).sum(1))) and np.array_equal(ref, d[idx])
    assert n_early >= 1 and n_fixed >= 5


def test_outliers_are_down_weighted():
    """What the robust mean is for: with a quarter of the track replaced by random descriptors the Cauchy mean stays closer to
    the inliers' direction than the plain mean does."""
    import pxo
    rng = np.random.default_rng(5)
    base = rng.normal(0, 1, 128); base /= np.linalg.norm(base)
    d = base + rng.normal(0, 0.03, (12, 128))
    d[:3] = rng.normal(0, 1, (3, 128))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    robust, _ = pxo.robust_mean_irls(d, pxo.loss("cauchy", 0.25), iters=100, l2_normalize=True)
    plain = d.mean(0); plain /= np.linalg.norm(plain)
    assert robust @ base > plain @ base and robust @ base > 0.99

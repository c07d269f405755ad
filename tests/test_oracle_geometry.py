"""CPU: oracle rows A6-A10, A19, A20 -- projection + camera models, residual functors, Ceres
loss/corrector, IRLS.  The reference has no golden vectors for these: they are
validated by finite differences, closed forms, and the reference's own property tests
(projection_test.cc:9-38 round trip, irls_optim_test.cc set-up)."""
import numpy as np
import pytest

import pxo

PARAMS = {0: [1200.0, 500, 480], 1: [1200.0, 1180, 500, 480], 2: [1200.0, 500, 480, 0.05],
          3: [1200.0, 500, 480, 0.05, -0.02], 4: [1200.0, 1180, 500, 480, 0.05, -0.02, 1e-3, -5e-4]}


def _fd(fun, x0, eps=1e-6):
    x0 = np.asarray(x0, dtype=np.float64)
    return np.stack([(fun(x0 + eps * np.eye(len(x0))[i]) - fun(x0 - eps * np.eye(len(x0))[i])) / (2 * eps)
                     for i in range(len(x0))], -1)


@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
def test_world_to_pixel_jacobians_finite_differences(model):
    rng = np.random.default_rng(model)
    k = np.array(PARAMS[model])
    for _ in range(5):
        q = rng.normal(size=4) * rng.uniform(0.5, 2.0)     # deliberately NOT unit norm
        t = rng.normal(size=3) + [0, 0, 6]
        X = rng.normal(size=3)
        xy, Jq, Jt, JX, Jk = pxo.world_to_pixel(model, k, q, t, X)
        f = lambda kk, qq, tt, XX: pxo.world_to_pixel(model, kk, qq, tt, XX, jac=False)[0]
        for J, fd in ((Jq, _fd(lambda z: f(k, z, t, X), q)), (Jt, _fd(lambda z: f(k, q, z, X), t)),
                      (JX, _fd(lambda z: f(k, q, t, z), X)), (Jk, _fd(lambda z: f(z, q, t, X), k, 1e-5))):
            assert np.abs(J - fd).max() < 1e-6 * max(1.0, np.abs(J).max())
        # ceres::QuaternionRotatePoint normalises q: the ambient Jacobian is orthogonal to q
        assert np.abs(Jq @ q).max() < 1e-9 * np.abs(Jq).max() * np.linalg.norm(q)
        # and the projection is invariant to the scale of q
        assert np.abs(f(k, 3.7 * q, t, X) - xy).max() < 1e-9


def test_projection_round_trip_like_reference_test():
    """projection_test.cc:9-38: identity pose, SIMPLE_PINHOLE and RADIAL; project, then undo by hand."""
    q, t = np.array([1.0, 0, 0, 0]), np.zeros(3)
    for model in (0, 3):
        k = np.array(PARAMS[model])
        for X in ([0.3, -0.2, 4.0], [1.0, 2.0, 10.0], [-0.5, 0.1, 2.5]):
            X = np.array(X)
            xy = pxo.world_to_pixel(model, k, q, t, X, jac=False)[0]
            u, v = X[0] / X[2], X[1] / X[2]
            if model == 0:
                want = np.array([k[0] * u + k[1], k[0] * v + k[2]])
            else:
                r2 = u * u + v * v
                rad = 1 + k[3] * r2 + k[4] * r2 * r2
                want = np.array([k[0] * u * rad + k[1], k[0] * v * rad + k[2]])
            assert np.abs(xy - want).max() < 1e-10


def test_rotation_matches_rotation_matrix():
    from pixsfm_amd import synthetic
    rng = np.random.default_rng(2)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    t, X = rng.normal(size=3) + [0, 0, 5], rng.normal(size=3)
    k = np.array(PARAMS[1])
    xy = pxo.world_to_pixel(1, k, q, t, X, jac=False)[0]
    p = synthetic.qvec_to_rotmat(q) @ X + t
    assert np.abs(xy - [k[0] * p[0] / p[2] + k[2], k[1] * p[1] / p[2] + k[3]]).max() < 1e-10


@pytest.mark.parametrize("name,a", [("trivial", 1.0), ("cauchy", 0.25), ("huber", 0.3), ("soft_l1", 0.5)])
def test_loss_functions_closed_form_and_derivatives(name, a):
    """[upstream Ceres loss_function.cc] rho, rho', rho'' -- closed forms + FD consistency; ScaledLoss."""
    ls = pxo.loss(name, a)
    b = a * a
    for s in (0.0, 1e-4, 0.05, 0.09, 0.5, 3.0):
        rho = pxo.loss_eval(ls, s)
        want = {"trivial": s, "cauchy": b * np.log1p(s / b),
                "huber": s if s <= b else 2 * a * np.sqrt(s) - b,
                "soft_l1": 2 * b * (np.sqrt(1 + s / b) - 1)}[name]
        assert abs(rho[0] - want) < 1e-14 * max(1, want)
        if s > 0 and not (name == "huber" and abs(s - b) < 1e-3):
            e = 1e-6 * max(s, 1e-3)
            assert abs(rho[1] - (pxo.loss_eval(ls, s + e)[0] - pxo.loss_eval(ls, s - e)[0]) / (2 * e)) < 1e-6
            assert abs(rho[2] - (pxo.loss_eval(ls, s + e)[1] - pxo.loss_eval(ls, s - e)[1]) / (2 * e)) < 1e-5 * max(1, abs(rho[2]))
        assert np.allclose(pxo.loss_eval(ls, s, weight=0.7), 0.7 * rho)


def test_corrector_gauss_newton_consistency():
    """[upstream corrector.cc] J~^T r~ = rho' J^T r and J~^T J~ = rho'(J^T J - kappa (J^T r)(J^T r)^T),
    the identities the GPU solver relies on (pxr_ba_solve.hip k_jac)."""
    rng = np.random.default_rng(0)
    r = rng.normal(size=128) * 0.05
    J = rng.normal(size=(128, 5))
    s = float(r @ r)
    for rho in (pxo.loss_eval(pxo.loss("cauchy", 0.25), s), np.array([0.3, 0.8, 0.4])):   # second: rho'' > 0 branch
        rt, Jt = pxo.corrector(s, rho, r, J)
        kappa = 0.0
        if rho[2] > 0:
            alpha = 1 - np.sqrt(1 + 2 * s * rho[2] / rho[1])
            kappa = (2 * alpha - alpha * alpha) / s
        b = J.T @ r
        assert np.allclose(Jt.T @ rt, rho[1] * b, rtol=1e-12, atol=1e-14)
        assert np.allclose(Jt.T @ Jt, rho[1] * (J.T @ J - kappa * np.outer(b, b)), rtol=1e-12, atol=1e-14)


def test_residual_functors_compose_projection_and_interpolation():
    """A7-A10: r and the 128 x n Jacobian blocks equal G * P assembled by hand."""
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=3, n_points=6, obs_per_point=2, seed=3, model=3, dtype=np.float64)
    i = 5
    img, pt = prob["obs_image"][i], prob["obs_point"][i]
    p = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
    k = prob["cam_params"][prob["image_camera"][img]][:5]
    q, t, X = prob["qvec"][img], prob["tvec"][img], prob["xyz"][pt]
    r, Jq, Jt, JX, Jk = pxo.ba_residual(p, pxo.cfg(), 3, q, t, X, k, prob["refs"][pt])
    xy, Pq, Pt, PX, Pk = pxo.world_to_pixel(3, k, q, t, X)
    f, gx, gy, _ = pxo.patch_eval(p, xy, pxo.cfg())
    assert np.allclose(r, f - prob["refs"][pt], atol=1e-15)
    for J, P in ((Jq, Pq), (Jt, Pt), (JX, PX), (Jk, Pk)):
        assert np.allclose(J, np.outer(gx, P[0]) + np.outer(gy, P[1]), atol=1e-13)
    # FD through the whole functor (fp64 patches)
    fd = _fd(lambda z: pxo.ba_residual(p, pxo.cfg(), 3, q, t, z, k, prob["refs"][pt], jac=False)[0], X, 1e-6)
    assert np.abs(fd - JX).max() < 1e-5 * max(1, np.abs(JX).max())
    # KA functor
    p2 = pxo.make_patch(prob["patches"][i - 1], prob["corners"][i - 1], prob["scales"][i - 1])
    kp1, kp2 = xy, prob["centers"][i - 1] + [0.3, -0.2]
    r, J1, J2 = pxo.ka_residual(p, p2, pxo.cfg(), kp1, kp2)
    f2, g2x, g2y, _ = pxo.patch_eval(p2, kp2, pxo.cfg())
    assert np.allclose(r, f - f2, atol=1e-15)
    assert np.allclose(J1, np.stack([gx, gy], 1)) and np.allclose(J2, -np.stack([g2x, g2y], 1))
    r8, J8 = pxo.ref2d_residual(p, pxo.cfg(), kp1, prob["refs"][pt])
    assert np.allclose(r8, f - prob["refs"][pt]) and np.allclose(J8, J1)


@pytest.mark.parametrize("C,n", [(128, 10), (128, 100), (3, 1000)])
def test_irls_robust_mean(C, n):
    """irls_optim_test.cc set-up (Cauchy(0.25), 100 iterations, l2_normalize = false).  Checked
    against a direct numpy transcription of base/src/irls_optim.h:43-71."""
    rng = np.random.default_rng(C + n)
    descs = rng.normal(0, 0.1, (n, C)) + rng.normal(0, 1, C)
    descs[: n // 10] += rng.normal(0, 2.0, (n // 10, C))       # outliers
    ls = pxo.loss("cauchy", 0.25)
    mean, early = pxo.robust_mean_irls(descs, ls, 100, l2_normalize=False)
    w = np.ones(n)
    want_early = -1
    for _ in range(100):
        w = w / w.sum()
        mu = (descs * w[:, None]).sum(0)
        s = ((descs - mu) ** 2).sum(1)
        rho0 = 0.0625 * np.log1p(s / 0.0625)
        if (rho0 <= 0).any():                       # irls_optim.h:63-66: return descriptor_track[i]
            want_early = int(np.argmax(rho0 <= 0))
            mu = descs[want_early]
            break
        w = 1.0 / rho0
    assert early == want_early and np.abs(mean - mu).max() < 1e-10
    inl = descs[n // 10:].mean(0)
    assert np.linalg.norm(mean - inl) < np.linalg.norm(descs.mean(0) - inl)     # outliers are down-weighted
    idx, ref, rm = pxo.compute_reference(descs, ls, 100, l2_normalize=False)
    assert idx == int(np.argmin(((descs - mean) ** 2).sum(1))) and np.array_equal(ref, descs[idx])


def test_irls_early_return_when_a_descriptor_equals_the_mean():
    d = np.tile(np.array([[0.6, 0.8, 0.0]]), (4, 1))
    mean, early = pxo.robust_mean_irls(d, pxo.loss("cauchy", 0.25), 100, l2_normalize=True)
    assert early == 0 and np.array_equal(mean, d[0])          # rho = 0 -> return descriptor_track[i], irls_optim.h:63-66

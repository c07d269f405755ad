"""The analytic Jacobians of the oracle against THIRD-PARTY automatic differentiation (torch.autograd, float64).

The reference obtains every Jacobian of the path by automatic differentiation (ceres::AutoDiffCostFunction over ceres::Jet,
residuals/src/feature_reference.h:38,91,182, featuremetric.h:39); the oracle and the HIP kernels carry hand-derived chain rules
instead (A3 normalisation, A4 J = G P, A6 projection).  Here the residual of a BA observation and of a KA edge is written once more
as a differentiable torch program -- Catmull-Rom bicubic with clamped indices (cubic_hermite_spline_simd.h:105-119, grid2d.h:64-73),
patch coordinates (featurepatch.h:250-255), L2 normalisation (interpolation.h:648-651), quaternion rotation with the normalisation
inside (ceres::QuaternionRotatePoint [upstream]), the COLMAP camera models [upstream] -- and differentiated by torch: an
independent autodiff engine on an independent transcription.  fp64 patches (all-fp64 arithmetic in the reference too), so the
two agree to rounding: values 1e-12, Jacobians 1e-9 relative."""
import numpy as np
import pytest
import torch

from cases import residual_cases as gen



def _cr(p0, p1, p2, p3, x):
    a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3)
    b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3)
    c = 0.5 * (-p0 + p2)
    return p1 + x * (c + x * (b + x * a))


def _bicubic(P, r, c):
    """P (H, W, C) tensor, r / c scalar tensors -> (C,) ; indices clamped to the patch (grid2d.h:64-73)"""
    H, W, _ = P.shape
    row, col = int(np.floor(r.item())), int(np.floor(c.item()))
    dy, dx = r - row, c - col
    rows = [min(max(row - 1 + j, 0), H - 1) for j in range(4)]
    cols = [min(max(col - 1 + i, 0), W - 1) for i in range(4)]
    h = [_cr(P[rows[j], cols[0]], P[rows[j], cols[1]], P[rows[j], cols[2]], P[rows[j], cols[3]], dx) for j in range(4)]
    return _cr(h[0], h[1], h[2], h[3], dy)


def _descriptor(P, corner, scale, xy, l2):
    u = xy[0] * scale[0] - 0.5 - corner[0]
    v = xy[1] * scale[1] - 0.5 - corner[1]
    f = _bicubic(P, v, u)
    return f / torch.linalg.norm(f) if l2 else f


def _rotate(q, X):
    q = q / torch.linalg.norm(q)
    w, x, y, z = q
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)]),
                     torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)]),
                     torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)])])
    return R @ X


def _world_to_image(model, k, u, v):
    """COLMAP 3.8 camera_models.h [upstream]: SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV"""
    r2 = u * u + v * v
    if model == 0:
        return k[0] * u + k[1], k[0] * v + k[2]
    if model == 1:
        return k[0] * u + k[2], k[1] * v + k[3]
    if model == 2:
        d = 1 + k[3] * r2
        return k[0] * u * d + k[1], k[0] * v * d + k[2]
    if model == 3:
        d = 1 + k[3] * r2 + k[4] * r2 * r2
        return k[0] * u * d + k[1], k[0] * v * d + k[2]
    rad = k[4] * r2 + k[5] * r2 * r2
    du = u * rad + 2 * k[6] * u * v + k[7] * (r2 + 2 * u * u)
    dv = v * rad + 2 * k[7] * u * v + k[6] * (r2 + 2 * v * v)
    return k[0] * (u + du) + k[2], k[1] * (v + dv) + k[3]


@pytest.mark.parametrize("l2", [True, False])
def test_ba_jacobian_equals_torch_autograd(l2):
    import pxo
    n = 0
    for c in gen.ba_cases():
        if c["l2"] != l2:
            continue
        d = c["d"].astype(np.float64)
        patch = pxo.make_patch(d, c["c"], c["s"])
        r, Jq, Jt, JX, Jk = pxo.ba_residual(patch, pxo.cfg(l2, False, False), c["model"], c["q"], c["t"], c["X"], c["params"], c["ref"])
        J = np.hstack([Jq, Jt, JX, Jk])
        xy0 = pxo.world_to_pixel(c["model"], c["params"], c["q"], c["t"], c["X"], jac=False)[0]
        uv = xy0 * c["s"] - 0.5 - c["c"]
        if np.abs(uv - np.round(uv)).min() < 1e-6:
            continue                                        # on a knot the one-sided derivatives differ: not a point autodiff defines
        P = torch.tensor(d)
        ref = torch.tensor(c["ref"])
        K = len(c["params"])

        def res(x):
            q, t, X, k = x[:4], x[4:7], x[7:10], x[10:]
            p = _rotate(q, X) + t
            px, py = _world_to_image(c["model"], k, p[0] / p[2], p[1] / p[2])
            return _descriptor(P, c["c"], c["s"], (px, py), l2) - ref
        x0 = torch.tensor(np.concatenate([c["q"], c["t"], c["X"], c["params"]]))
        r_t = res(x0).numpy()
        J_t = torch.autograd.functional.jacobian(res, x0).numpy()
        assert J_t.shape == (128, 10 + K)
        assert np.abs(r_t - r).max() < 1e-12, (c["name"], np.abs(r_t - r).max())
        assert np.abs(J_t - J).max() < 1e-9 * max(1.0, np.abs(J).max()), (c["name"], np.abs(J_t - J).max(), np.abs(J).max())
        n += 1
    assert n >= 8


def test_ka_edge_jacobian_equals_torch_autograd():
    import pxo
    n = 0
    for c in gen.ka_cases():
        if c["float_simd"]:
            continue
        d1, d2 = c["d1"].astype(np.float64), c["d2"].astype(np.float64)
        p1, p2 = pxo.make_patch(d1, c["c1"], c["s1"]), pxo.make_patch(d2, c["c2"], c["s2"])
        r, J1, J2 = pxo.ka_residual(p1, p2, pxo.cfg(c["l2"], False, False), c["kp1"], c["kp2"])
        uv = np.concatenate([c["kp1"] * c["s1"] - 0.5 - c["c1"], c["kp2"] * c["s2"] - 0.5 - c["c2"]])
        if np.abs(uv - np.round(uv)).min() < 1e-6:
            continue
        P1, P2 = torch.tensor(d1), torch.tensor(d2)

        def res(x):
            return _descriptor(P1, c["c1"], c["s1"], (x[0], x[1]), c["l2"]) - _descriptor(P2, c["c2"], c["s2"], (x[2], x[3]), c["l2"])
        x0 = torch.tensor(np.concatenate([c["kp1"], c["kp2"]]))
        J_t = torch.autograd.functional.jacobian(res, x0).numpy()
        assert np.abs(res(x0).numpy() - r).max() < 1e-12
        J = np.hstack([J1, J2])
        assert np.abs(J_t - J).max() < 1e-9 * max(1.0, np.abs(J).max()), (c["name"], np.abs(J_t - J).max())
        n += 1
    assert n >= 20

"""CPU: the cost-map oracle (oracle/pxo_costmap.py) against what can be pinned of the reference.

The reference has no test or golden vector for CostMapExtractor, so:
  * the two storage-type rounding rules the extraction leans on are pinned bit-exactly against the reference's
    OWN vendored half.hpp compiled from its own source (oracle/ref_half_shim.cc -> oracle/_ref/libpxo_ref_half.so): half - half, and dtype(double) of SetEntry;
  * the vectorised restatement is checked against an independent per-texel loop that follows
    costmap_extractor.h:242-357 statement by statement;
  * the 3-channel, reference-free residual block (costmap_bundle_optimizer.h:104-119) is checked by finite differences.
"""
import ctypes as C

import numpy as np
import pytest

import pxo
import pxo_costmap


def _ref():
    r = pxo.ref()
    if r is None or not hasattr(r, "pxo_ref_half_sub"):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    return r


def test_half_subtraction_rule_matches_reference_half_hpp():
    r = _ref()
    rng = np.random.default_rng(1)
    n = 300000
    a = (rng.normal(size=n) * 10.0 ** rng.uniform(-7, 4, n)).astype(np.float16)
    b = (rng.normal(size=n) * 10.0 ** rng.uniform(-7, 4, n)).astype(np.float16)
    b[:1000] = a[:1000]                                    # exact cancellation
    out = np.empty_like(a)
    r.pxo_ref_half_sub(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(n))
    mine = pxo_costmap.storage_diff(a, b)
    same = (out.view(np.uint16) == mine.view(np.uint16)) | ((out == 0) & (mine == 0))
    assert same.all()


def test_set_entry_cast_rule_matches_reference_half_hpp():
    """dtype(value) for half goes through float (half.hpp has no constructor from double): two roundings.  A single
    correct rounding (numpy's float64 -> float16) differs on ~3e-5 of random inputs -- the test would catch it."""
    r = _ref()
    rng = np.random.default_rng(2)
    v = rng.normal(size=400000) * 10.0 ** rng.uniform(-9, 4, 400000)
    h = rng.normal(size=2000).astype(np.float16)
    ties = h.astype(np.float64) + np.spacing(h).astype(np.float64) / 2       # exact half-way points and their neighbours
    v = np.concatenate([v, ties, ties * (1 + 2e-16), ties * (1 - 2e-16), ties * (1 + 1e-9)])
    out = np.empty(v.size, np.uint16)
    r.pxo_ref_half_from_double(v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(v.size))
    assert np.array_equal(out, pxo_costmap.store(v, np.float16).view(np.uint16))
    assert (out != v.astype(np.float16).view(np.uint16)).any()


def test_half_rules_match_the_committed_golden_vectors():
    """tests/golden/half_rules.npz was produced by the reference's half.hpp (tests/golden/make_golden_half.py); it pins
    the oracle's two rounding rules wherever oracle/_ref cannot be built."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "half_rules.npz"))
    a, b = g["a"].view(np.float16), g["b"].view(np.float16)
    mine = pxo_costmap.storage_diff(a, b).view(np.uint16)
    want = g["sub"]
    zero = (mine & 0x7fff == 0) & (want & 0x7fff == 0)                 # +0 / -0 of exact cancellations
    assert ((mine == want) | zero).all()
    assert np.array_equal(pxo_costmap.store(g["v"], np.float16).view(np.uint16), g["cast"])


def _loop_costmap(patch, ref, loss, as_gradientfield, apply_sqrt):
    """costmap_extractor.h:242-357 one statement at a time (doubles), no vectorisation."""
    H, W, Cn = patch.shape
    ls = pxo.loss(*loss)
    out = np.zeros((H, W, 3 if as_gradientfield else 1))
    for y in range(H):
        for x in range(W):
            f = patch[y, x].astype(np.float64)
            res = f - ref
            rho = pxo.loss_eval(ls, float(res @ res))
            cost = 0.5 * rho[0]
            if not as_gradientfield:
                out[y, x, 0] = np.sqrt(cost) if apply_sqrt else cost
                continue
            top, bottom = min(H - 1, y + 1), max(0, y - 1)
            right, left = min(W - 1, x + 1), max(0, x - 1)
            dfdr = 0.5 * (patch[top, x] - patch[bottom, x]).astype(np.float64)       # numpy subtracts in the storage type
            dfdc = 0.5 * (patch[y, right] - patch[y, left]).astype(np.float64)
            dr = dc = 0.0
            if cost > 1e-8:
                dr, dc = rho[1] * (res @ dfdr), rho[1] * (res @ dfdc)
                if apply_sqrt:
                    cost = np.sqrt(cost)
                    dr *= 0.5 / cost
                    dc *= 0.5 / cost
            out[y, x] = cost, dr, dc
    return out


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
@pytest.mark.parametrize("kw", [dict(), dict(loss=("cauchy", 0.25)), dict(apply_sqrt=True, loss=("cauchy", 0.25)),
                                dict(as_gradientfield=False), dict(as_gradientfield=False, apply_sqrt=True)])
def test_vectorised_costmap_equals_the_statement_by_statement_loop(dtype, kw):
    rng = np.random.default_rng(3)
    patch = rng.normal(size=(7, 9, 16))
    patch /= np.linalg.norm(patch, axis=-1, keepdims=True)
    patch = patch.astype(dtype)
    ref = patch[3, 4].astype(np.float64)           # one texel with zero cost (the `cost > 1e-8` branch)
    full = dict(loss=("trivial", 1.0), as_gradientfield=True, apply_sqrt=False)
    full.update(kw)
    want = _loop_costmap(patch, ref, **full)
    got = pxo_costmap.fill_point_costmap(patch, ref, out_dtype=np.float64, **kw)
    assert np.abs(got - want).max() < 1e-14
    assert np.all(got[3, 4] == 0.0)
    # stored in the patch's dtype by default (bundle_adjustment/bindings.cc:21)
    assert pxo_costmap.fill_point_costmap(patch, ref, **kw).dtype == dtype


def test_cost_channel_gradient_channels_are_consistent():
    """On a smooth field the derivative channels approximate the finite differences of the cost channel (what
    `as_gradientfield` is for): same central-difference stencil, chain rule through rho."""
    yy, xx = np.meshgrid(np.arange(16.0), np.arange(16.0), indexing="ij")
    k = np.arange(1, 9)[None, None, :]
    patch = np.concatenate([np.sin(0.02 * k * xx[..., None] + 0.03 * yy[..., None]), np.cos(0.02 * k * yy[..., None] - 0.02 * xx[..., None])], -1)
    ref = patch[8, 8] + 0.3
    m = pxo_costmap.fill_point_costmap(patch, ref, loss=("cauchy", 0.5))
    fd_r = 0.5 * (m[2:, 1:-1, 0] - m[:-2, 1:-1, 0])
    fd_c = 0.5 * (m[1:-1, 2:, 0] - m[1:-1, :-2, 0])
    assert np.abs(fd_r - m[1:-1, 1:-1, 1]).max() < 0.05 * np.abs(m[..., 1]).max()
    assert np.abs(fd_c - m[1:-1, 1:-1, 2]).max() < 0.05 * np.abs(m[..., 2]).max()


@pytest.mark.parametrize("channels", [3, 1])
def test_reference_free_residual_block_finite_differences(channels):
    """FeatureReferenceCostFunctor with ref_descriptor = nullptr on a few-channel patch ([upstream] scalar Ceres bicubic,
    interpolation.h:222-268): residual = interpolated texel, Jacobians by the chain rule -- checked numerically."""
    rng = np.random.default_rng(4)
    data = rng.normal(size=(16, 16, channels))
    patch = pxo.make_patch(data, corner=(492, 492))
    q = np.array([0.99, 0.05, -0.03, 0.02]); t = np.array([0.1, -0.2, 5.0]); X = np.array([0.05, -0.04, 0.3])
    params = np.array([1200.0, 500.0, 500.0, 0.02])
    cfg = pxo.cfg(l2_normalize=False)
    r, Jq, Jt, JX, Jk = pxo.ba_residual(patch, cfg, 2, q, t, X, params, None)
    assert r.shape == (channels,)
    # the value is the plain bicubic texel at the projection
    xy = pxo.world_to_pixel(2, params, q, t, X, jac=False)[0]
    f = pxo.patch_eval(patch, xy, cfg, want_grad=False)[0]
    assert np.array_equal(r, f)
    h = 1e-6
    for J, base, idx in ((JX, X, 0), (Jt, t, 1), (Jk, params, 2), (Jq, q, 3)):
        for j in range(len(base)):
            args = [X.copy(), t.copy(), params.copy(), q.copy()]
            args[idx][j] += h
            rp = pxo.ba_residual(patch, cfg, 2, args[3], args[1], args[0], args[2], None, jac=False)[0]
            args[idx][j] -= 2 * h
            rm = pxo.ba_residual(patch, cfg, 2, args[3], args[1], args[0], args[2], None, jac=False)[0]
            fd = (rp - rm) / (2 * h)
            assert np.abs(fd - J[:, j]).max() < 1e-5 * max(1.0, np.abs(J[:, j]).max())


def test_cross_derivative_and_interpolated_branch_by_finite_differences():
    """The oracle's cross derivative d2f/drdc (the fourth output of PixelInterpolator::Evaluate, interpolation.h:642-646;
    not touched by the L2 normalisation) and the cost-map cross term (costmap_extractor.h:304-308) against central
    differences of the quantities they differentiate."""
    import pxo
    import pxo_costmap
    rng = np.random.default_rng(8)
    patch = rng.normal(0, 1, (8, 8, 16)).astype(np.float64)
    P = pxo.make_patch(patch)
    cfg = pxo.cfg(l2_normalize=False)
    h = 1e-5
    for r, c in ((3.3, 4.7), (2.05, 2.9), (5.5, 1.25)):
        f, dr, dc, drc = pxo.pixel_interp_cross(P, r, c, cfg)
        fd = (pxo.pixel_interp_cross(P, r, c + h, cfg)[1] - pxo.pixel_interp_cross(P, r, c - h, cfg)[1]) / (2 * h)
        assert np.abs(drc - fd).max() < 1e-6 * max(1.0, np.abs(fd).max())
        f2, dr2, dc2 = pxo.pixel_interp(P, r, c, cfg)
        assert np.array_equal(f, f2) and np.array_equal(dr, dr2) and np.array_equal(dc, dc2)
    # cost-map cross term: d/dc of dcost/dr, evaluated through an upsampled grid (texel spacing 1/64)
    ref = rng.normal(0, 1, 16)
    up = 64.0
    cm = pxo_costmap.fill_point_costmap_interpolated(patch[:3, :3].copy(), ref, cfg, loss=("cauchy", 0.5), upsampling_factor=up,
                                                     compute_cross_derivative=True, out_dtype=np.float64)
    assert cm.shape == (192, 192, 4)
    y, x = 70, 90
    fd_r = (cm[y + 1, x, 0] - cm[y - 1, x, 0]) * up / 2            # dcost/dr by differences of the cost channel
    fd_c = (cm[y, x + 1, 0] - cm[y, x - 1, 0]) * up / 2
    fd_rc = (cm[y, x + 1, 1] - cm[y, x - 1, 1]) * up / 2           # d/dc of the dcost/dr channel
    assert abs(cm[y, x, 1] - fd_r) < 1e-3 * max(1.0, abs(fd_r)) and abs(cm[y, x, 2] - fd_c) < 1e-3 * max(1.0, abs(fd_c))
    assert abs(cm[y, x, 3] - fd_rc) < 2e-3 * max(1.0, abs(fd_rc))
    assert pxo_costmap.cost_patch_shape(16, 16, 1.5, 3) == (24, 24, 3) and pxo_costmap.cost_patch_shape(12, 12, 1.0, 1) == (12, 12, 1)

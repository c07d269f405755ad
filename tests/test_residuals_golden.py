"""The residual functors of the hot path (SURVEY 8a rows A7-A10 and the Jet bridge A4) pinned against the REFERENCE's own
code: tests/golden/residuals_ref.npz holds residuals and Jacobians of pixsfm's FeatureMetric2DCostFunctor (KA edge),
FeatureReference2DCostFunctor (unary reference term), FeatureReferenceCostFunctor and FeatureReferenceConstantPoseCostFunctor
(BA) -- residuals/src/featuremetric.h, feature_reference.h, base/src/projection.h compiled in place and differentiated with
one dual number per parameter like ceres::AutoDiffCostFunction (tests/golden/make_golden_residuals.py,
oracle/ref_residual_shim.cc).  Underneath the functors the quaternion rotation and the camera models are stubs restated from
the published Ceres / COLMAP definitions, so what is pinned is the functors' composition, the interpolation stack and the
chain rule through them -- the camera models themselves (row A6) stay "vs the published formulas".
Checked here: the oracle's C restatement on the CPU, the HIP kernels (pxr_ka_eval, pxr_ba_eval + pxr_ba_projection_jacobian)
on the GPU."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_residuals", os.path.join(HERE, "golden", "make_golden_residuals.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _rel(a, b):
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


def _gold():
    return np.load(os.path.join(HERE, "golden", "residuals_ref.npz"))


def test_oracle_ka_and_unary_residuals_match_the_reference_vectors():
    import pxo
    gen, gold = _gen(), _gold()
    for c in gen.ka_cases():
        n = c["name"]
        cfg = pxo.cfg(c["l2"], c["float_simd"], False)
        p1, p2 = pxo.make_patch(c["d1"], c["c1"], c["s1"]), pxo.make_patch(c["d2"], c["c2"], c["s2"])
        r, J1, J2 = pxo.ka_residual(p1, p2, cfg, c["kp1"], c["kp2"])
        assert _rel(r, gold[n + "_r"]) < 1e-13 and _rel(J1, gold[n + "_J1"]) < 1e-13 and _rel(J2, gold[n + "_J2"]) < 1e-13, n
        r2, Jk = pxo.ref2d_residual(p1, cfg, c["kp1"], c["ref"])
        assert _rel(r2, gold[n + "_r2d"]) < 1e-13 and _rel(Jk, gold[n + "_J2d"]) < 1e-13, n


def test_oracle_ba_residuals_match_the_reference_vectors():
    import pxo
    gen, gold = _gen(), _gold()
    for c in gen.ba_cases():
        n = c["name"]
        patch = pxo.make_patch(c["d"], c["c"], c["s"])
        r, Jq, Jt, JX, Jk = pxo.ba_residual(patch, pxo.cfg(c["l2"], False, c["check_bounds"]), c["model"], c["q"], c["t"], c["X"],
                                            c["params"], c["ref"])
        J = np.hstack([Jq, Jt, JX, Jk])
        assert J.shape == gold[n + "_J"].shape
        assert _rel(r, gold[n + "_r"]) < 1e-12, (n, _rel(r, gold[n + "_r"]))
        # columns qvec | tvec | point | camera parameters; the quaternion block is the derivative THROUGH the normalisation
        # of QuaternionRotatePoint (4 columns, not a tangent-space 3)
        assert _rel(J, gold[n + "_J"]) < 1e-10, (n, _rel(J, gold[n + "_J"]))
        for blk, sl in (("q", slice(0, 4)), ("t", slice(4, 7)), ("X", slice(7, 10)), ("k", slice(10, None))):
            assert _rel(J[:, sl], gold[n + "_J"][:, sl]) < 1e-9, (n, blk)


def test_reference_run_live_when_present():
    gen, gold = _gen(), _gold()
    if not os.path.exists(gen.LIB):
        pytest.skip("oracle/_ref/libpxo_ref_residual.so not built (reference tree absent)")
    for c in gen.ka_cases()[:6]:
        r, J1, J2, r2, Jk = gen.run_ka(c)
        assert np.array_equal(r, gold[c["name"] + "_r"]) and np.array_equal(J1, gold[c["name"] + "_J1"])
    for c in gen.ba_cases()[:10]:
        r, J, ok = gen.run_ba(c, False)
        assert np.array_equal(r, gold[c["name"] + "_r"]) and np.array_equal(J, gold[c["name"] + "_J"])
        rc, Jc, _ = gen.run_ba(c, True)            # the constant-pose functor: the point / camera columns of the same Jacobian
        assert np.array_equal(rc, r) and np.array_equal(Jc, J[:, 7:])


@pytest.mark.gpu
def test_hip_ka_edges_match_the_reference_vectors():
    from pixsfm_amd.engine import Context, PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    gen, gold = _gen(), _gold()
    ctx = Context(0)
    groups = {}
    for c in gen.ka_cases():
        groups.setdefault((c["d1"].dtype, c["l2"], c["float_simd"]), []).append(c)
    n_checked = 0
    for (dt, l2, fs), cs in groups.items():
        m = len(cs)
        prob = dict(kp=np.concatenate([[c["kp1"], c["kp2"]] for c in cs]), node_patch=np.arange(2 * m, dtype=np.int64),
                    node_const=np.zeros(2 * m, np.uint8), node_problem=np.zeros(2 * m, np.int32),
                    edge_src=np.arange(0, 2 * m, 2, dtype=np.int32), edge_dst=np.arange(1, 2 * m, 2, dtype=np.int32),
                    edge_w=np.ones(m), patches=np.concatenate([[c["d1"], c["d2"]] for c in cs]),
                    corners=np.concatenate([[c["c1"], c["c2"]] for c in cs]).astype(np.int32),
                    scales=np.concatenate([[c["s1"], c["s2"]] for c in cs]), n_problems=1)
        arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
        ka = KAProblem(ctx, arena, prob)
        cost, r, J1, J2 = ka.eval(interp_cfg(l2_normalize=l2, use_float_simd=fs), make_loss("trivial", []), materialize=True)
        r, J1, J2 = r.download(), J1.download(), J2.download()
        tol = 1e-9 if fs else 1e-10
        for i, c in enumerate(cs):
            n = c["name"]
            assert _rel(r[i], gold[n + "_r"]) < tol and _rel(J1[i], gold[n + "_J1"]) < tol and _rel(J2[i], gold[n + "_J2"]) < tol, n
            n_checked += 1
    assert n_checked == len(gen.ka_cases())


@pytest.mark.gpu
def test_hip_ba_residuals_match_the_reference_vectors():
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg
    gen, gold = _gen(), _gold()
    ctx = Context(0)
    groups = {}
    for c in gen.ba_cases():
        groups.setdefault((c["l2"], c["check_bounds"]), []).append(c)
    n_checked = 0
    for (l2, cb), cs in groups.items():
        m = len(cs)
        cam_params = np.zeros((m, 12))
        for i, c in enumerate(cs):
            cam_params[i, :len(c["params"])] = c["params"]
        ids = np.arange(m, dtype=np.int32)
        prob = dict(obs_image=ids, obs_point=ids, obs_patch=np.arange(m, dtype=np.int64), image_camera=ids,
                    qvec=np.stack([c["q"] for c in cs]), tvec=np.stack([c["t"] for c in cs]),
                    cam_model=np.array([c["model"] for c in cs], np.int32), cam_params=cam_params,
                    xyz=np.stack([c["X"] for c in cs]), refs=np.stack([c["ref"] for c in cs]),
                    patches=np.stack([c["d"] for c in cs]), corners=np.stack([c["c"] for c in cs]).astype(np.int32),
                    scales=np.stack([c["s"] for c in cs]))
        arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
        ba = BAProblem(ctx, arena, prob)
        rec, r, gx, gy = ba.eval(interp_cfg(l2_normalize=l2, check_bounds=cb), with_jacobian=True, materialize=True)
        P = ba.projection_jacobian().download()
        r, gx, gy = r.download(), gx.download(), gy.download()
        J = gx[:, :, None] * P[:, None, 0, :] + gy[:, :, None] * P[:, None, 1, :]          # C x 22: q | t | X | 12 camera slots
        for i, c in enumerate(cs):
            n = c["name"]
            K = len(c["params"])
            assert _rel(r[i], gold[n + "_r"]) < 1e-10, (n, _rel(r[i], gold[n + "_r"]))
            assert _rel(J[i][:, :10 + K], gold[n + "_J"]) < 1e-9, (n, _rel(J[i][:, :10 + K], gold[n + "_J"]))
            assert np.all(J[i][:, 10 + K:] == 0.0)
            n_checked += 1
    assert n_checked == len(gen.ba_cases())


@pytest.mark.gpu
def test_single_block_cost_functions_of_the_shim_match_the_reference_vectors():
    """`_pixsfm._residuals.FeatureReferenceCostFunctor / FeatureReferenceConstantPoseCostFunctor` (residuals/bindings.cc:14-30):
    one residual block with the ceres::CostFunction surface, residuals and per-block Jacobians against the vectors of the
    reference's functors.  The first factory ignores its reference descriptor like the reference's binding does
    (feature_reference.h:267-271), so its residual is the golden residual PLUS the reference."""
    from pixsfm_amd._pixsfm import _residuals
    from pixsfm_amd.api import features
    from pixsfm_amd.engine import Context
    gen, gold = _gen(), _gold()
    ctx = Context(0)
    n = 0
    for c in gen.ba_cases():
        if c["check_bounds"] or c["d"].shape[2] != 128:
            continue
        name = c["name"]
        K = len(c["params"])
        patch = features.FeaturePatch(c["d"], c["c"], c["s"])
        icfg = {"l2_normalize": c["l2"]}
        f = _residuals.FeatureReferenceCostFunctor(c["model"], patch, c["ref"].reshape(1, -1), icfg, ctx=ctx)
        assert f.num_residuals() == 128 and f.parameter_block_sizes() == [4, 3, 3, K]
        ok, r, J = f.evaluate(c["q"], c["t"], c["X"], c["params"])
        assert ok and _rel(r, gold[name + "_r"] + c["ref"]) < 1e-10
        assert [b.shape for b in J] == [(128, 4), (128, 3), (128, 3), (128, K)]
        assert _rel(np.hstack(J), gold[name + "_J"]) < 1e-9
        g = _residuals.FeatureReferenceConstantPoseCostFunctor(c["model"], c["q"], c["t"], patch, c["ref"].reshape(1, -1), icfg, ctx=ctx)
        assert g.parameter_block_sizes() == [3, K]
        ok, r, J = g.evaluate(c["X"], c["params"])
        assert ok and _rel(r, gold[name + "_r"]) < 1e-10 and _rel(np.hstack(J), gold[name + "_J"][:, 7:]) < 1e-9
        n += 1
        if n >= 12:
            break
    assert n >= 6
    with pytest.raises(ValueError, match="Unsupported dimensions"):
        _residuals.FeatureReferenceCostFunctor(0, features.FeaturePatch(np.zeros((4, 4, 64), np.float16), (0, 0), (1.0, 1.0)), np.zeros((1, 64)), {}, ctx=ctx)
    with pytest.raises(NotImplementedError):
        _residuals.GeometricCostFunctor(0, np.zeros(2))

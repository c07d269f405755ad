"""GPU parity: fused BA residual kernel (pxr_ba_eval, through the C-ABI) vs the CPU oracle.

Tolerance: BASELINE.json's north_star asks residuals/Jacobians within 1e-5 relative; the
kernel keeps the reference's precision contract (fp32 horizontal / fp64 vertical), so we
assert 1e-10 relative for the default mode and for use_float_simd.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _setup(ctx, **kw):
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena
    prob = synthetic.make_ba_problem(**kw)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    return prob, arena, BAProblem(ctx, arena, prob)


def _oracle_blocks(prob, cfg_kw):
    import pxo
    cost, r, J = pxo.ba_eval_batch(prob, pxo.cfg(**cfg_kw), pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    return cost, r, J


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("float_simd", [False, True])
def test_residual_and_jacobian_match_oracle(ctx, model, float_simd):
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ba = _setup(ctx, n_cams=5, n_points=67, obs_per_point=3, seed=10 + model, model=model)
    rec, r, gx, gy = ba.eval(interp_cfg(use_float_simd=float_simd), with_jacobian=True, materialize=True)
    P = ba.projection_jacobian().download()
    r, gx, gy, rec = r.download(), gx.download(), gy.download(), rec.download()
    cost_o, r_o, J_o = _oracle_blocks(prob, dict(use_float_simd=float_simd))
    assert _relerr(r, r_o) < TOL
    # J = [gx gy] * P  (C x 22)
    J = gx[:, :, None] * P[:, None, 0, :] + gy[:, :, None] * P[:, None, 1, :]
    assert _relerr(J, J_o) < TOL
    # fused record == reductions of the materialised quantities
    assert _relerr(rec[:, 0], (r_o ** 2).sum(1)) < TOL
    assert _relerr(rec[:, 1], (gx * gx).sum(1)) < TOL
    assert _relerr(rec[:, 2], (gx * gy).sum(1)) < 1e-8
    assert _relerr(rec[:, 3], (gy * gy).sum(1)) < TOL
    assert np.abs(rec[:, 4] - (gx * r_o).sum(1)).max() < 1e-9 * np.abs(gx).max()
    assert np.abs(rec[:, 5] - (gy * r_o).sum(1)).max() < 1e-9 * np.abs(gy).max()
    cost = ba.cost(make_loss("cauchy", [0.25]))
    assert abs(cost - cost_o) < 1e-10 * abs(cost_o)


@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
@pytest.mark.parametrize("channels", [128, 64])
def test_patch_dtypes_and_channels(ctx, dtype, channels):
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _setup(ctx, n_cams=4, n_points=33, obs_per_point=3, seed=3, dtype=dtype, channels=channels)
    rec, r, gx, gy = ba.eval(interp_cfg(), with_jacobian=True, materialize=True)
    cost_o, r_o, J_o = _oracle_blocks(prob, {})
    P = ba.projection_jacobian().download()
    J = gx.download()[:, :, None] * P[:, None, 0, :] + gy.download()[:, :, None] * P[:, None, 1, :]
    assert _relerr(r.download(), r_o) < TOL
    assert _relerr(J, J_o) < TOL


def test_scaled_patches_border_clamp_and_ragged_tail(ctx):
    """scale != 1 (featurepatch.h:250-255), projections pushed against / beyond the patch border
    (Grid2D clamping, grid2d.h:64-73), n_obs not a multiple of the 16-observation lane group."""
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _setup(ctx, n_cams=6, n_points=101, obs_per_point=5, seed=5, scale=(0.5, 0.25),
                             rot_deg=1.5, trans=0.05, pt_sigma=0.05)
    assert ba.n_obs % 16 != 0
    rec, r, gx, gy = ba.eval(interp_cfg(), with_jacobian=True, materialize=True)
    cost_o, r_o, J_o = _oracle_blocks(prob, {})
    uv = (rec.download()[:, 6:8] * prob["scales"] - 0.5 - prob["corners"])
    assert (uv.min() < 1.0) or (uv.max() > 14.0), "test should exercise border clamping"
    assert _relerr(r.download(), r_o) < TOL
    P = ba.projection_jacobian().download()
    J = gx.download()[:, :, None] * P[:, None, 0, :] + gy.download()[:, :, None] * P[:, None, 1, :]
    assert _relerr(J, J_o) < TOL


def test_value_only_and_unnormalised(ctx):
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _setup(ctx, n_cams=4, n_points=40, obs_per_point=3, seed=8)
    rec, r, _, _ = ba.eval(interp_cfg(l2_normalize=False), with_jacobian=False, materialize=True)
    cost_o, r_o, _ = _oracle_blocks(prob, dict(l2_normalize=False))
    assert _relerr(r.download(), r_o) < TOL
    assert _relerr(rec.download()[:, 0], (r_o ** 2).sum(1)) < TOL


def test_empty_problem_is_a_noop(ctx):
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, make_loss
    prob = synthetic.make_ba_problem(n_cams=3, n_points=4, obs_per_point=2, seed=1)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    for k in ("obs_image", "obs_point", "obs_patch"):
        prob[k] = prob[k][:0]
    ba = BAProblem(ctx, arena, prob)
    ba.eval(interp_cfg())
    assert ba.cost(make_loss()) == 0.0


def test_unsupported_channels_raise(ctx):
    from pixsfm_amd import PixsfmHipError
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _setup(ctx, n_cams=3, n_points=5, obs_per_point=2, seed=1, channels=24)
    with pytest.raises(PixsfmHipError):
        ba.eval(interp_cfg())


@pytest.mark.parametrize("name,dtype,fs", [("f16", np.float16, 0), ("f16", np.float16, 1), ("f32", np.float32, 0),
                                           ("f64", np.float64, 0)])
def test_kernel_reproduces_the_oracles_bare_bicubic(ctx, name, dtype, fs):
    """The bare bicubic (no normalisation) at the positions of tests/cases/bicubic_cases.py (the shape of the reference's
    TestBiCubicSimilarCeres, interpolation_test.cc:327-364: border, outside, on-texel).  Drive the fused kernel so that it
    evaluates exactly those (r, c): identity pose, SIMPLE_PINHOLE f = 1, c = 0, X = (c + .5, r + .5, 1), zero reference =>
    r = f, gx = dfdc, gy = dfdr; compare with the oracle (same fp32-horizontal / fp64-vertical split, same FMA order)."""
    import pxo
    from cases import bicubic_cases
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg
    pos = bicubic_cases.positions()
    n = len(pos)
    data = np.ascontiguousarray(bicubic_cases.grid(name).astype(dtype))
    grid = data[None]
    prob = dict(obs_image=np.zeros(n, np.int32), obs_point=np.arange(n, dtype=np.int32),
                obs_patch=np.zeros(n, np.int64), image_camera=np.zeros(1, np.int32),
                qvec=np.array([[1.0, 0, 0, 0]]), tvec=np.zeros((1, 3)), cam_model=np.zeros(1, np.int32),
                cam_params=np.array([[1.0, 0.0, 0.0]]), refs=np.zeros((n, 128)),
                xyz=np.stack([pos[:, 1] + 0.5, pos[:, 0] + 0.5, np.ones(n)], 1))
    arena = PatchArena.from_numpy(ctx, grid, np.zeros((1, 2), np.int32), np.ones((1, 2)))
    ba = BAProblem(ctx, arena, prob)
    _, r, gx, gy = ba.eval(interp_cfg(l2_normalize=False, use_float_simd=bool(fs)), with_jacobian=True, materialize=True)
    p = pxo.make_patch(data)
    want = np.stack([np.stack(pxo.bicubic(p, float(rr), float(cc), bool(fs))) for rr, cc in pos])
    tol = 1e-12 * np.abs(want).max()          # (c + .5) - .5 may differ from c by one ulp
    assert np.abs(r.download() - want[:, 0]).max() < tol
    assert np.abs(gy.download() - want[:, 1]).max() < tol
    assert np.abs(gx.download() - want[:, 2]).max() < tol


def test_linearity_in_reference_full_size_property(ctx):
    """Size-independent property: r(ref) - r(0) == -ref for every observation."""
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _setup(ctx, n_cams=8, n_points=500, obs_per_point=4, seed=21)
    _, r1, _, _ = ba.eval(interp_cfg(), with_jacobian=False, materialize=True)
    r1 = r1.download()
    ba.d["refs"].upload(np.zeros_like(prob["refs"]))
    rec0, r0, _, _ = ba.eval(interp_cfg(), with_jacobian=False, materialize=True)
    r0 = r0.download()
    assert np.abs((r0 - r1) - prob["refs"][prob["obs_point"]]).max() < 1e-15
    assert np.abs((r0 ** 2).sum(1) - 1.0).max() < 1e-12   # unit-norm descriptors (interpolation_test.cc:187-207)


def test_check_bounds_has_no_effect_with_reference_descriptors(ctx):
    """InterpolationConfig.check_bounds: PatchInterpolator::Evaluate reports whether the projection lies inside its patch
    (patch_interpolator.h:125-135,160-166), but FeatureReferenceCostFunctor passes that on only when it has NO reference
    descriptor (`if (!ref_descriptor_) return is_inside; ... return true;`, feature_reference.h:128-136 -- the cost-map
    functor, tests/test_costmap_gpu.py::test_costmap_ba_check_bounds).  With references the evaluation is the border-clamped
    one regardless: records, cost and solve are those of check_bounds = False, and equal the oracle's."""
    import pxo
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=4, n_points=40, obs_per_point=3, seed=3)
    prob["corners"] = prob["corners"].copy()
    prob["corners"][::5, 0] += 9                      # shift some patches: the projection falls left of the patch
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    rec_on = ba.eval(interp_cfg(check_bounds=True), with_jacobian=True)[0].download()
    rec_off = ba.eval(interp_cfg(check_bounds=False), with_jacobian=True)[0].download()
    inside = []
    for i in range(len(prob["obs_image"])):
        p = pxo.make_patch(prob["patches"][i], prob["corners"][i], prob["scales"][i])
        inside.append(pxo.patch_eval(p, rec_off[i, 6:8], pxo.cfg(check_bounds=True))[3] == 1)
    assert 0 < (~np.array(inside)).sum() < len(inside)            # some observations ARE outside their patches
    assert np.isfinite(rec_on).all() and np.array_equal(rec_on, rec_off)
    cost_o, _, _ = pxo.ba_eval_batch(prob, pxo.cfg(check_bounds=True), pxo.loss("cauchy", 0.25))
    cost = ba.cost(make_loss("cauchy", [0.25]))
    assert abs(cost - cost_o) < 1e-10 * cost_o
    gauge = (np.array([1, 0, 0, 0], np.uint8), np.array([0, 1, 0, 0], np.uint8), np.full(4, 0b0110, np.uint16),
             np.zeros(40, np.uint8))
    for inner in (False, True):
        for name in ("qvec", "tvec", "xyz", "cam_params"):
            ba.d[name].upload(prob[name])
        s_on = ba.solve(interp_cfg(check_bounds=True), make_loss("cauchy", [0.25]), *gauge,
                        options=lm_options(max_iterations=3, use_inner_iterations=inner))
        p_on = [a.copy() for a in ba.params()]
        for name in ("qvec", "tvec", "xyz", "cam_params"):
            ba.d[name].upload(prob[name])
        s_off = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=3, use_inner_iterations=inner))
        assert s_on["termination"] != 2 and s_on["iterations"] == s_off["iterations"], (s_on, s_off)
        # (the two solves are separate runs: their reductions agree to rounding, not bit for bit)
        assert abs(s_on["final_cost"] - s_off["final_cost"]) <= 1e-9 * abs(s_off["final_cost"]), (s_on, s_off)
        assert all(np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()) for a, b in zip(p_on, ba.params()))


@pytest.mark.parametrize("wild", [3e9, 1e12, 1e30, float("inf"), float("nan")])
def test_projections_beyond_int_range_stay_inside_the_patch(ctx, wild):
    """A diverging camera can project observations billions of texels away from their patch.  The texel index is clamped
    to the border like Grid2D does (grid2d.h:64-73) BEFORE the double -> int conversion (undefined beyond int's range:
    the kernel used to fault there); the observations of the other cameras are not disturbed."""
    from pixsfm_amd.engine import interp_cfg
    prob, arena, ba = _setup(ctx, n_cams=6, n_points=120, obs_per_point=4, seed=12)
    rec0, r0, gx0, _ = ba.eval(interp_cfg(), with_jacobian=True, materialize=True)
    rec0, r0, gx0 = rec0.download(), r0.download(), gx0.download()
    cams = np.zeros((6, 12))
    cams[:, :prob["cam_params"].shape[1]] = prob["cam_params"]
    cams[::2, 0] = wild                                    # focal length of every other camera
    ba.d["cam_params"].upload(cams)
    rec, r, gx, _ = ba.eval(interp_cfg(), with_jacobian=True, materialize=True)
    ctx.sync()
    rec, r, gx = rec.download(), r.download(), gx.download()
    hit = prob["image_camera"][prob["obs_image"]] % 2 == 0
    assert hit.any() and (~hit).any()
    np.testing.assert_array_equal(r[~hit], r0[~hit])
    np.testing.assert_array_equal(gx[~hit], gx0[~hit])
    if np.isfinite(wild):
        assert np.isfinite(r[hit]).all()                   # border texels, normalised

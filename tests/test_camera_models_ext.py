"""The six less common COLMAP models of CAMERA_MODEL_SWITCH_CASES (feature_reference.h:232):
OPENCV_FISHEYE 5, FULL_OPENCV 6, FOV 7, SIMPLE_RADIAL_FISHEYE 8, RADIAL_FISHEYE 9,
THIN_PRISM_FISHEYE 10.  CPU: the oracle (complex-step Jacobians) vs finite differences and vs
closed-form limits.  GPU: residuals / Jacobians of the fused kernel (hand-written templated
formulas + forward-mode duals) vs the oracle, and a short LM solve."""
import numpy as np
import pytest

import pxo

EXT = {5: [1200.0, 1180, 500, 480, 0.02, -0.01, 0.003, -0.001],
       6: [1200.0, 1180, 500, 480, 0.05, -0.02, 1e-3, -5e-4, 0.01, 0.02, -0.01, 0.005],
       7: [1200.0, 1180, 500, 480, 0.9],
       8: [1200.0, 500, 480, 0.03],
       9: [1200.0, 500, 480, 0.03, -0.01],
       10: [1200.0, 1180, 500, 480, 0.03, -0.01, 1e-3, -5e-4, 0.004, -0.002, 1e-3, 2e-3]}


def _fd(fun, x0, eps):
    x0 = np.asarray(x0, dtype=np.float64)
    return np.stack([(fun(x0 + eps * np.eye(len(x0))[i]) - fun(x0 - eps * np.eye(len(x0))[i])) / (2 * eps)
                     for i in range(len(x0))], -1)


@pytest.mark.parametrize("model", sorted(EXT))
def test_oracle_complex_step_jacobians(model):
    rng = np.random.default_rng(model)
    k = np.array(EXT[model])
    assert pxo.lib().pxo_camera_num_params(model) == len(k)
    for _ in range(4):
        q = rng.normal(size=4)
        t = rng.normal(size=3) * 0.2 + [0, 0, 4]
        X = rng.normal(size=3) * 0.6
        xy, Jq, Jt, JX, Jk = pxo.world_to_pixel(model, k, q, t, X)
        f = lambda kk, qq, tt, XX: pxo.world_to_pixel(model, kk, qq, tt, XX, jac=False)[0]
        for J, fd in ((Jq, _fd(lambda z: f(k, z, t, X), q, 1e-6)), (Jt, _fd(lambda z: f(k, q, z, X), t, 1e-6)),
                      (JX, _fd(lambda z: f(k, q, t, z), X, 1e-6)), (Jk, _fd(lambda z: f(z, q, t, X), k, 1e-6))):
            assert np.abs(J - fd).max() < 2e-6 * max(1.0, np.abs(J).max())


def test_model_limits():
    """zero distortion: every model reduces to the pinhole projection; FOV's three branches agree
    across their switch-over points (omega^2 = 1e-4, radius^2 = 1e-4)."""
    u, v = 0.21, -0.13
    for model, k in ((6, [900.0, 950, 400, 300] + [0.0] * 8),):
        xy, _, _ = pxo.world_to_image(model, np.array(k), u, v)
        assert np.abs(xy - [900 * u + 400, 950 * v + 300]).max() < 1e-12
    # fisheye family with zero coefficients: x = f * theta * u / r
    r = np.hypot(u, v); th = np.arctan(r)
    for model, k in ((5, [900.0, 950, 400, 300, 0, 0, 0, 0]), (10, [900.0, 950, 400, 300] + [0.0] * 8)):
        xy, _, _ = pxo.world_to_image(model, np.array(k), u, v)
        assert np.abs(xy - [900 * th * u / r + 400, 950 * th * v / r + 300]).max() < 1e-12
    for model, k in ((8, [900.0, 400, 300, 0.0]), (9, [900.0, 400, 300, 0.0, 0.0])):
        xy, _, _ = pxo.world_to_image(model, np.array(k), u, v)
        assert np.abs(xy - [900 * th * u / r + 400, 900 * th * v / r + 300]).max() < 1e-12
    for om in (0.0099, 0.0101):          # omega^2 just below / above 1e-4
        a, _, _ = pxo.world_to_image(7, np.array([900.0, 950, 400, 300, om]), u, v)
        want = np.arctan(2 * r * np.tan(om / 2)) / (r * om)
        assert np.abs(a - [900 * u * want + 400, 950 * v * want + 300]).max() < 1e-4   # 3rd-order Taylor branch
    for uu in (0.0099, 0.0101):          # radius^2 just below / above 1e-4
        a, _, _ = pxo.world_to_image(7, np.array([900.0, 950, 400, 300, 0.8]), uu, 0.0)
        want = np.arctan(2 * uu * np.tan(0.4)) / (uu * 0.8)
        assert abs(a[0] - (900 * uu * want + 400)) < 1e-5


def _problem(model, seed):
    from pixsfm_amd import synthetic
    base = 0 if model in (8, 9) else 1                 # single / double focal layout
    prob = synthetic.make_ba_problem(n_cams=5, n_points=48, obs_per_point=3, seed=seed, model=base, rot_deg=0.1)
    k = np.array(EXT[model], dtype=np.float64)
    small = k.copy()
    if model == 7:
        small[4] = 0.02                                 # FOV: mild, third branch for most radii
    else:
        nf = 3 if base == 0 else 4
        small[nf:] *= 0.05                              # keep the projections inside the rendered patches
        small[:nf] = prob["cam_params"][0, :nf]
    prob["cam_model"] = np.full(len(prob["cam_model"]), model, dtype=np.int32)
    cp = np.zeros((len(prob["cam_model"]), 12)); cp[:, :len(small)] = small
    prob["cam_params"] = cp
    return prob


@pytest.mark.gpu
@pytest.mark.parametrize("model", sorted(EXT))
def test_gpu_residuals_and_jacobians_match_oracle(ctx, model):
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg
    prob = _problem(model, seed=30 + model)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    rec, r, gx, gy = ba.eval(interp_cfg(), with_jacobian=True, materialize=True)
    P = ba.projection_jacobian().download()
    _, r_o, J_o = pxo.ba_eval_batch(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), want_r=True, want_J=True)
    J = gx.download()[:, :, None] * P[:, None, 0, :] + gy.download()[:, :, None] * P[:, None, 1, :]
    assert np.abs(r.download() - r_o).max() < 1e-10 * np.abs(r_o).max()
    assert np.abs(J - J_o).max() < 1e-9 * np.abs(J_o).max()


@pytest.mark.gpu
@pytest.mark.parametrize("evaluation", ["texels", "gram"])
@pytest.mark.parametrize("model,inner", [(5, False), (7, False), (10, False), (5, True), (10, True)])   # FOV + inner iterations: see below
def test_gpu_lm_with_extended_models_matches_oracle(ctx, exact_ctx, model, inner, evaluation):
    """evaluation "texels": the exact-order kernel at every candidate (the reference's fp32 pass: what the oracle restates, tight
    tolerances); "gram": the default evaluation from cached Gram matrices (conftest.FP32_PASS_*; the ill-conditioned FOV scene,
    whose trajectory amplifies any rounding, is compared on the exact-order path only)."""
    from conftest import FP32_PASS_FINAL_COST_RTOL, FP32_PASS_PARAM_RTOL
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    if evaluation == "texels":
        ctx = exact_ctx
    elif model == 7:
        pytest.skip("ill-conditioned FOV scene: compared on the exact-order path")
    prob = _problem(model, seed=60 + model)
    n_img = len(prob["image_camera"])
    K = len(EXT[model])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    cmask = np.full(n_img, 0b1100 if K > 4 and model != 8 else 0b0110, np.uint16)   # keep the principal point
    if model == 7:
        cmask[:] = 0b11100      # FOV: omega sits next to the omega^2 = 1e-4 branch switch -> keep it fixed
    gauge = (pose_const, tmask, cmask, np.zeros(48, np.uint8))
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    # inner: the nested per-point LM (csrc/pxr_ba_inner.hip) takes the models' d(x,y)/d(u,v) without d(x,y)/dk
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=5, use_inner_iterations=inner))
    q, t, k, X = ba.params()
    so, qo, to, ko, Xo = pxo.ba_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), *gauge,
                                      pxo.lm_options(max_iterations=5, use_inner_iterations=int(inner)))
    # (FOV with the nested LMs on top is left out: which of the five steps are accepted on this ill-conditioned scene depends on
    # the summation order of the atomics and changes from run to run)
    assert s["iterations"] == so["iterations"] and s["num_successful"] == so["num_successful"]
    if model == 7:
        # FOV couples focal length and depth almost degenerately on this scene: the LM trajectories are
        # ill-conditioned, so only the (matching) Jacobians above and the cost level are compared
        assert abs(s["final_cost"] - so["final_cost"]) < 1e-2 * so["final_cost"]
        return
    tol = 1e-4 if inner else 1e-6           # the nested LMs stop on 1e-6 relative tolerances: test_ba_inner_gpu.py
    if evaluation == "gram":
        tol = max(tol, FP32_PASS_FINAL_COST_RTOL, FP32_PASS_PARAM_RTOL)
    assert abs(s["final_cost"] - so["final_cost"]) < tol * max(so["final_cost"], 1e-9)
    assert np.abs(q - qo).max() < tol and np.abs(X - Xo).max() < tol and np.abs(k - ko).max() < 10 * tol * 1200

"""GPU: the iterative Schur solver of pxr_ba_solve (the reference's > 1000-image regime: ITERATIVE_SCHUR +
SCHUR_JACOBI, bundle_adjustment/src/bundle_optimizer.h:180-191).  Ceres itself cannot be run here, so the checks are
(a) against the direct solver (exact Schur complement + dense Cholesky, itself tested against the oracle) on problems
where both fit -- with the conjugate gradients driven to a tight residual the two must take the same LM steps -- and
(b) the behaviour of the reference's default inexact configuration (eta = 0.1, at most 200 linear iterations)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gauge(n_img, n_cam, n_pts, cam_mask=0b0110):
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    return pose_const, tmask, np.full(n_cam, cam_mask, np.uint16), np.zeros(n_pts, np.uint8)


def _solve(ctx, arena, prob, gauge, **opts):
    from pixsfm_amd.engine import BAProblem, interp_cfg, lm_options, make_loss
    ba = BAProblem(ctx, arena, prob)
    s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(**opts))
    return s, ba.params()


def _close(pa, pb, tol):
    for a_, b_ in zip(pa, pb):
        assert np.abs(a_ - b_).max() < tol * max(1.0, np.abs(b_).max()), np.abs(a_ - b_).max()


@pytest.mark.parametrize("shared_camera", [False, True])
def test_tight_cg_takes_the_direct_solvers_steps(ctx, shared_camera):
    """One camera per image (joint pose + intrinsics preconditioner blocks) and ONE camera shared by all images
    (pose blocks + one intrinsics block)."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import PatchArena
    n_img, n_pts = 24, 600
    prob = synthetic.make_ba_problem(n_cams=n_img, n_points=n_pts, obs_per_point=4, seed=5, shared_camera=shared_camera)
    gauge = _gauge(n_img, len(prob["cam_model"]), n_pts)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    tight = dict(linear_solver="iterative", eta=0.0, linear_r_tolerance=1e-13, max_linear_solver_iterations=2000)
    for iters in (1, 4):          # (at convergence the accept / reject decision of a step is rounding noise)
        sd, pd = _solve(ctx, arena, prob, gauge, max_iterations=iters, linear_solver="direct")
        si, pi = _solve(ctx, arena, prob, gauge, max_iterations=iters, **tight)
        assert sd["linear_solver"] == 1 and si["linear_solver"] == 2 and si["linear_iterations"] > 0
        assert si["iterations"] == sd["iterations"] and si["num_successful"] == sd["num_successful"]
        assert abs(si["final_cost"] - sd["final_cost"]) < 1e-8 * sd["initial_cost"]
        _close(pi, pd, 1e-7)


def test_reference_default_inexact_steps_converge(ctx):
    """eta = 0.1 / 200 linear iterations (bundle_adjustment_options.h:55): inexact Newton steps -- a different
    trajectory from the direct solver's, the same optimum."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import PatchArena
    n_img, n_pts = 30, 800
    prob = synthetic.make_ba_problem(n_cams=n_img, n_points=n_pts, obs_per_point=5, seed=11)
    gauge = _gauge(n_img, n_img, n_pts)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    sd, pd = _solve(ctx, arena, prob, gauge, max_iterations=25, linear_solver="direct")
    si, pi = _solve(ctx, arena, prob, gauge, max_iterations=25, linear_solver="iterative")
    assert si["final_cost"] < 0.01 * si["initial_cost"]
    assert abs(si["final_cost"] - sd["final_cost"]) < 1e-3 * sd["final_cost"] + 1e-9 * sd["initial_cost"]
    assert 0 < si["linear_iterations"] <= 200 * si["iterations"]
    _close(pi, pd, 1e-4)                                    # the 1e-4 pose / point bar of the metric, vs our direct solver


def test_auto_selection_follows_the_image_count(ctx):
    """<= 1000 images: direct; more: iterative (kMaxNumImagesDirectSparseSolver, bundle_optimizer.h:179-190).
    1001 images with three observations each is enough to see the switch."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import PatchArena
    out = {}
    for n_img in (1000, 1001):
        n_pts = 700
        prob = synthetic.make_ba_problem(n_cams=n_img, n_points=n_pts, obs_per_point=5, seed=3, channels=64, patch_size=8)
        gauge = _gauge(n_img, n_img, n_pts)
        arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
        s, _ = _solve(ctx, arena, prob, gauge, max_iterations=2)
        out[n_img] = s
        arena.close()
    assert out[1000]["linear_solver"] == 1 and out[1001]["linear_solver"] == 2
    assert out[1001]["final_cost"] < out[1001]["initial_cost"]


def test_constant_points_and_constant_cameras(ctx):
    """Gauge variants: some points constant, intrinsics fully constant (pose-only blocks), poses constant for a few
    images (their columns vanish)."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import PatchArena
    n_img, n_pts = 16, 300
    prob = synthetic.make_ba_problem(n_cams=n_img, n_points=n_pts, obs_per_point=4, seed=21)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    pose_const, tmask, cmask, ptc = _gauge(n_img, n_img, n_pts, cam_mask=0b1111)
    pose_const[[3, 7]] = 1
    ptc[::7] = 1
    gauge = (pose_const, tmask, cmask, ptc)
    tight = dict(linear_solver="iterative", eta=0.0, linear_r_tolerance=1e-13, max_linear_solver_iterations=2000)
    sd, pd = _solve(ctx, arena, prob, gauge, max_iterations=5, linear_solver="direct")
    si, pi = _solve(ctx, arena, prob, gauge, max_iterations=5, **tight)
    assert si["num_camera_unknowns"] == sd["num_camera_unknowns"] == 6 * (n_img - 3) - 1
    assert abs(si["final_cost"] - sd["final_cost"]) < 1e-8 * sd["initial_cost"]
    _close(pi, pd, 1e-7)


def test_two_thousand_cameras_against_the_direct_solver(ctx):
    """2000 cameras (16k camera unknowns: the direct solver's [S | rhs] is 2 GB, still affordable) -- the regime the
    reference hands to ITERATIVE_SCHUR.  Tight CG reproduces the direct LM steps; the default configuration reaches the
    same cost level."""
    import torch
    from pixsfm_amd import synthetic_gpu
    from pixsfm_amd.engine import PatchArena
    n_img, n_pts, opp = 2000, 40_000, 5
    prob, patches = synthetic_gpu.make_ba_problem_gpu("cuda:0", n_cams=n_img, n_points=n_pts, obs_per_point=opp, seed=9,
                                                      channels=128, patch_size=16)
    arena = PatchArena(ctx, len(prob["obs_image"]), 16, 16, 128, np.float16, device_ptr=patches.data_ptr())
    arena.upload(0, None, prob["corners"], prob["scales"])
    gauge = _gauge(n_img, n_img, n_pts)
    s_auto, p_auto = _solve(ctx, arena, prob, gauge, max_iterations=3)
    assert s_auto["linear_solver"] == 2 and s_auto["num_camera_unknowns"] == 8 * n_img - 7
    sd, pd = _solve(ctx, arena, prob, gauge, max_iterations=3, linear_solver="direct")
    si, pi = _solve(ctx, arena, prob, gauge, max_iterations=3, linear_solver="iterative", eta=0.0,
                    linear_r_tolerance=1e-12, max_linear_solver_iterations=3000)
    assert si["iterations"] == sd["iterations"] and si["num_successful"] == sd["num_successful"]
    assert abs(si["final_cost"] - sd["final_cost"]) < 1e-7 * sd["initial_cost"]
    _close(pi, pd, 1e-6)
    assert s_auto["final_cost"] < 1e-3 * s_auto["initial_cost"]       # inexact steps: slower per iteration, same direction
    del arena, patches
    torch.cuda.empty_cache()


def test_aachen_shaped_scene_at_a_tenth_of_the_size(ctx):
    """BASELINE configs[4] (Aachen Day-Night scale: ~4000 images, >= 5M observations, 8 x 8 fp16 patches like the reference's
    low_memory.yaml:7) at a tenth of the size -- 400 cameras, 100k points, 500k observations: the iterative solver the
    reference would pick above 1000 images, driven tight, against the direct solver (3193 x 3193 Cholesky) on the same
    scene, and the default inexact configuration; then the cost-map strategy on the same patches (`bench.py --preset
    aachen` runs the full size)."""
    import torch
    from pixsfm_amd import synthetic_gpu
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    n_img, n_pts, opp = 400, 100_000, 5
    # 8 x 8 patches leave +-2 px around the 4 x 4 stencil: initial errors of a pixel, not the four of the 16 x 16 scenes
    prob, patches = synthetic_gpu.make_ba_problem_gpu("cuda:0", n_cams=n_img, n_points=n_pts, obs_per_point=opp, seed=4,
                                                      channels=128, patch_size=8, rot_deg=0.04, trans=0.003, pt_sigma=0.003)
    arena = PatchArena(ctx, len(prob["obs_image"]), 8, 8, 128, np.float16, device_ptr=patches.data_ptr())
    arena.upload(0, None, prob["corners"], prob["scales"])
    gauge = _gauge(n_img, n_img, n_pts)
    sd, pd = _solve(ctx, arena, prob, gauge, max_iterations=3, linear_solver="direct")
    si, pi = _solve(ctx, arena, prob, gauge, max_iterations=3, linear_solver="iterative", eta=0.0,
                    linear_r_tolerance=1e-12, max_linear_solver_iterations=3000)
    assert sd["num_camera_unknowns"] == 8 * n_img - 7 and sd["num_successful"] >= 2
    assert si["iterations"] == sd["iterations"] and si["num_successful"] == sd["num_successful"]
    assert abs(si["final_cost"] - sd["final_cost"]) < 1e-7 * sd["initial_cost"]
    _close(pi, pd, 1e-6)
    s_def, _ = _solve(ctx, arena, prob, gauge, max_iterations=8, linear_solver="iterative")
    assert s_def["linear_iterations"] > 0 and s_def["final_cost"] < 1e-2 * s_def["initial_cost"]
    # the low-memory strategy on the same scene: 3-channel cost maps from the 8 x 8 patches, BA on the maps
    ba = BAProblem(ctx, arena, prob)
    cm = ba.extract_costmaps(make_loss("trivial", []))
    cba = ba.costmap_problem(cm)
    s_cm = cba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), *gauge,
                     options=lm_options(max_iterations=8, linear_solver="iterative"))
    assert s_cm["num_successful"] >= 3 and s_cm["final_cost"] < 0.5 * s_cm["initial_cost"]
    del ba, cba, cm, arena, patches
    torch.cuda.empty_cache()

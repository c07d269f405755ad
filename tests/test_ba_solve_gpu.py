"""GPU parity: pxr_ba_solve (Schur LM on the GPU) vs the oracle's dense LM (oracle/pxo_solve.c).

north_star tolerance: refined poses/points within 1e-4 on identical inputs.  Both solvers
restate the same Ceres trust-region loop, so trajectories agree far tighter; we assert 1e-6
on parameters and 1e-8 relative on the cost trajectory end points.
PARITY UNPINNED w.r.t. real Ceres (not available): the oracle is the comparison target.
"""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gauge(prob, refine_focal=True, refine_pp=False, refine_extra=True):
    """default_problem_setup (pixsfm/bundle_adjustment/main.py:12-18) + BundleOptimizerOptions defaults."""
    n_img, n_cam, n_pt = len(prob["image_camera"]), len(prob["cam_model"]), len(prob["xyz"])
    pose_const = np.zeros(n_img, np.uint8); pose_const[0] = 1
    tmask = np.zeros(n_img, np.uint8); tmask[1] = 1
    # parameter index groups [upstream COLMAP]: focal / principal point / extra
    groups = {0: ([0], [1, 2], []), 1: ([0, 1], [2, 3], []), 2: ([0], [1, 2], [3]), 3: ([0], [1, 2], [3, 4]),
              4: ([0, 1], [2, 3], [4, 5, 6, 7])}
    cmask = np.zeros(n_cam, np.uint16)
    for c, m in enumerate(prob["cam_model"]):
        f, pp, ex = groups[int(m)]
        const = ([] if refine_focal else f) + ([] if refine_pp else pp) + ([] if refine_extra else ex)
        cmask[c] = sum(1 << a for a in const)
    return pose_const, tmask, cmask, np.zeros(n_pt, np.uint8)


@pytest.fixture(params=["texels", "gram"])
def ctx(request, exact_ctx):
    """Every comparison with the oracle runs twice: on the exact-order evaluation (the reference's arithmetic: tight tolerances)
    and on the default one (cached Gram matrices: the tolerances of conftest.FP32_PASS_*)."""
    if request.param == "texels":
        return exact_ctx
    return request.getfixturevalue("default_ctx")


@pytest.fixture(scope="module")
def default_ctx():
    from pixsfm_amd.engine import Context
    c = Context(0)
    assert c.gram_cache and c.deterministic
    yield c
    c.close()


def _run_both(ctx, prob, gauge, max_it=25, **opt_kw):
    import pxo
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    s_gpu = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=max_it, **opt_kw))
    q, t, k, X = ba.params()
    s_cpu, qo, to, ko, Xo = pxo.ba_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), *gauge,
                                         pxo.lm_options(max_iterations=max_it, **opt_kw))
    s_gpu["_gram"] = bool(ctx.gram_cache)
    return s_gpu, (q, t, k, X), s_cpu, (qo, to, ko, Xo)


def _assert_same(s_gpu, pg, s_cpu, po, ptol=1e-6, trajectory=True):
    """trajectory=True: while the solve is still descending both LM loops must take identical
    accept/reject decisions.  Once the cost sits on the fp16 quantisation floor (~1e-5) the
    decisions hinge on 1e-17 cost differences, so converged runs compare end points only."""
    if trajectory:
        assert s_gpu["iterations"] == s_cpu["iterations"]
        assert s_gpu["num_successful"] == s_cpu["num_successful"]
        assert s_gpu["termination"] == s_cpu["termination"]
    from conftest import FP32_PASS_FINAL_COST_RTOL, FP32_PASS_PARAM_RTOL
    ctol = 1e-6
    if s_gpu.get("_gram"):       # the default evaluation differs from the oracle's by the fp32 pass's rounding (conftest.py)
        ctol, ptol = FP32_PASS_FINAL_COST_RTOL, max(ptol, FP32_PASS_PARAM_RTOL)
    assert abs(s_gpu["initial_cost"] - s_cpu["initial_cost"]) < 1e-10 * s_cpu["initial_cost"]
    assert abs(s_gpu["final_cost"] - s_cpu["final_cost"]) < ctol * max(s_cpu["final_cost"], 1e-9)
    for a, b, name in zip(pg, po, ("qvec", "tvec", "cam", "xyz")):
        b = np.asarray(b)
        a = a[:, :b.shape[1]] if a.ndim == 2 else a
        assert np.abs(a - b).max() < ptol * max(1.0, np.abs(b).max()), name


@pytest.mark.parametrize("model", [2, 0, 4])
def test_default_gauge_matches_oracle(ctx, model):
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=6, n_points=80, obs_per_point=4, seed=40 + model, model=model)
    res = _run_both(ctx, prob, _gauge(prob), max_it=7)
    _assert_same(*res)
    res = _run_both(ctx, prob, _gauge(prob), max_it=30)
    _assert_same(*res, ptol=1e-4, trajectory=False)               # north_star: poses/points within 1e-4
    s_gpu = res[0]
    assert s_gpu["final_cost"] < 1e-3 * s_gpu["initial_cost"]     # converges to the rendered optimum


def test_shared_intrinsics_constant_points_and_subsets(ctx):
    """one camera shared by all images (dense intrinsics column), some constant points, principal
    point refined, extra params constant, a second constant pose."""
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=7, n_points=90, obs_per_point=5, seed=77, model=3, shared_camera=True)
    pose_const, tmask, cmask, ptc = _gauge(prob, refine_focal=True, refine_pp=True, refine_extra=False)
    pose_const[3] = 1
    tmask[2] = 0b101
    ptc[::7] = 1
    res = _run_both(ctx, prob, (pose_const, tmask, cmask, ptc), max_it=6)
    _assert_same(*res)
    q, t, k, X = res[1]
    assert np.array_equal(X[::7], prob["xyz"][::7])               # constant points untouched
    assert np.array_equal(t[3], prob["tvec"][3])
    assert t[2][0] == prob["tvec"][2][0] and t[2][2] == prob["tvec"][2][2]
    assert np.array_equal(k[0, 3:5], prob["cam_params"][0, 3:5])  # extra params constant


def test_points_only_and_tolerances(ctx):
    """all poses/cameras constant (n_c = 0: pure point refinement) + function tolerance termination."""
    from pixsfm_amd import synthetic
    prob = synthetic.make_ba_problem(n_cams=5, n_points=60, obs_per_point=4, seed=5, rot_deg=0.0, trans=0.0)
    n_img = len(prob["image_camera"])
    gauge = (np.ones(n_img, np.uint8), np.zeros(n_img, np.uint8), np.full(n_img, 0xF, np.uint16), np.zeros(60, np.uint8))
    res = _run_both(ctx, prob, gauge, max_it=50, function_tolerance=1e-4)
    _assert_same(*res)
    assert res[0]["termination"] == 0 and res[0]["iterations"] < 50


def test_two_rank_partition_on_one_gpu(ctx):
    """The multi-GPU path (points sharded, cameras replicated, all-reduce of the reduced camera
    system) exercised with two solver instances on ONE GPU: two host threads, an in-process
    sum as the all-reduce.  Must reproduce the single-rank solution."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.parallel import shard_ba_problem
    prob = synthetic.make_ba_problem(n_cams=6, n_points=64, obs_per_point=4, seed=91)
    gauge = _gauge(prob)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    s_ref = ba.solve(interp_cfg(), make_loss(), *gauge, options=lm_options(max_iterations=6))
    ref = ba.params()

    world = 2
    barrier = threading.Barrier(world)
    stage = {}
    results = [None] * world

    def worker(rank):
        c = Context(0)
        c.gram_cache, c.deterministic = ctx.gram_cache, ctx.deterministic          # the arithmetic of the one-rank solve
        shard, pt_ids = shard_ba_problem(prob, rank, world)
        a = PatchArena.from_numpy(c, shard["patches"], shard["corners"], shard["scales"])
        b = BAProblem(c, a, shard)
        g = (gauge[0], gauge[1], gauge[2], gauge[3][pt_ids])

        def allreduce(ptr, count):
            import ctypes as C
            buf = np.empty(count)
            c.sync()
            c.lib.pxr_memcpy_d2h(c.handle, buf.ctypes.data, C.c_void_p(ptr), count * 8)
            stage[rank] = buf
            barrier.wait()
            tot = stage[0] + stage[1]
            barrier.wait()
            c.lib.pxr_memcpy_h2d(c.handle, C.c_void_p(ptr), tot.ctypes.data, count * 8)

        s = b.solve(interp_cfg(), make_loss(), *g, options=lm_options(max_iterations=6), allreduce=allreduce)
        results[rank] = (s, b.params(), pt_ids)
        c.sync()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert all(r is not None for r in results)
    for s, (q, t, k, X), pt_ids in results:
        assert s["iterations"] == s_ref["iterations"] and s["num_successful"] == s_ref["num_successful"]
        assert abs(s["final_cost"] - s_ref["final_cost"]) < 1e-9 * max(1e-9, s_ref["final_cost"]) + 1e-14
        assert np.abs(q - ref[0]).max() < 1e-9 and np.abs(t - ref[1]).max() < 1e-9
        assert np.abs(k - ref[2]).max() < 1e-7
        assert np.abs(X - ref[3][pt_ids]).max() < 1e-9
        if ctx.deterministic:          # integer sums, all-reduced as exact 32-bit halves through this sum-of-doubles callback: the same bits
            assert s["final_cost"] == s_ref["final_cost"] and np.array_equal(q, ref[0]) and np.array_equal(X, ref[3][pt_ids])


def test_device_built_observation_lists_equal_the_host_ones(ctx, monkeypatch):
    """Set-up: the per-image / per-point observation lists are built on the device (stable counting sort by image) when the
    observations are ordered by point; PXR_BA_SETUP_HOST=1 forces the host construction, an input that is NOT ordered by
    point takes it by itself.  Same lists -> the same summation orders -> the solves agree to the reduction noise of
    two separate runs, and a permuted input reproduces the ordered one."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=9, n_points=700, obs_per_point=5, seed=123, shared_camera=True)
    gauge = _gauge(prob)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])

    def solve(p):
        ba = BAProblem(ctx, arena, p)
        s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *gauge, options=lm_options(max_iterations=4, use_inner_iterations=True))
        return s, ba.params()

    s_dev, p_dev = solve(prob)
    monkeypatch.setenv("PXR_BA_SETUP_HOST", "1")
    s_host, p_host = solve(prob)
    monkeypatch.delenv("PXR_BA_SETUP_HOST")
    perm = np.random.default_rng(5).permutation(len(prob["obs_image"]))      # not ordered by point any more
    shuffled = dict(prob, obs_image=prob["obs_image"][perm], obs_point=prob["obs_point"][perm], obs_patch=prob["obs_patch"][perm])
    s_perm, p_perm = solve(shuffled)
    for s_, p_ in ((s_host, p_host), (s_perm, p_perm)):
        assert s_["iterations"] == s_dev["iterations"] and s_["num_successful"] == s_dev["num_successful"]
        assert abs(s_["final_cost"] - s_dev["final_cost"]) <= 1e-9 * abs(s_dev["final_cost"])
        assert all(np.abs(a - b).max() <= 1e-9 * max(1.0, np.abs(b).max()) for a, b in zip(p_, p_dev))


@pytest.mark.parametrize("channels,inner", [(3, False), (3, True), (1, False)])
def test_feature_reference_ba_on_image_intensities(ctx, channels, inner):
    """dense_features.model.name = "image": 3- or 1-channel maps, references subtracted, no L2 normalisation --
    FeatureReferenceBundleOptimizer's (3, 1) and (1, 1) cases (feature_reference_bundle_optimizer.h:13-16).  Same LM
    trajectory as the oracle; the point-only inner iterations run on the few-channel kernel with references."""
    import pxo
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, PatchArena, interp_cfg, lm_options, make_loss
    prob = synthetic.make_ba_problem(n_cams=6, n_points=60, obs_per_point=4, seed=70 + channels, channels=channels, noise=0.02)
    gauge = _gauge(prob)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ba = BAProblem(ctx, arena, prob)
    kw = dict(max_iterations=6, use_inner_iterations=inner)
    s_gpu = ba.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25]), *gauge, options=lm_options(**kw))
    s_cpu, qo, to, ko, Xo = pxo.ba_solve(prob, pxo.cfg(l2_normalize=False), pxo.loss("cauchy", 0.25), *gauge, pxo.lm_options(**kw))
    _assert_same(s_gpu, ba.params(), s_cpu, (qo, to, ko, Xo))
    assert s_gpu["final_cost"] < s_gpu["initial_cost"]


@pytest.mark.parametrize("inner", [False, True])
def test_work_buffer_arena_changes_nothing_and_follows_the_problem_size(monkeypatch, inner):
    """Round 6: the solve's work buffers come out of one grow-only allocation of the context, sized by the PREVIOUS solve.  A small
    problem, a larger one (the arena is too small: per-buffer allocation for what does not fit, then it grows), the small one again,
    and the same three with PXR_BA_ARENA=0: identical bits everywhere (deterministic default)."""
    from pixsfm_amd import synthetic
    from pixsfm_amd.engine import BAProblem, Context, PatchArena, interp_cfg, lm_options, make_loss
    probs = [synthetic.make_ba_problem(n_cams=4, n_points=40, obs_per_point=3, seed=3),
             synthetic.make_ba_problem(n_cams=7, n_points=260, obs_per_point=4, seed=4),
             synthetic.make_ba_problem(n_cams=4, n_points=40, obs_per_point=3, seed=3)]

    def run(knob):
        if knob is None:
            monkeypatch.delenv("PXR_BA_ARENA", raising=False)
        else:
            monkeypatch.setenv("PXR_BA_ARENA", knob)
        c = Context(0)
        out = []
        for prob in probs:
            arena = PatchArena.from_numpy(c, prob["patches"], prob["corners"], prob["scales"])
            ba = BAProblem(c, arena, prob)
            s = ba.solve(interp_cfg(), make_loss("cauchy", [0.25]), *_gauge(prob),
                         options=lm_options(max_iterations=8, use_inner_iterations=inner))
            out.append((s["final_cost"], s["iterations"], s["num_successful"]) + ba.params())
            arena.close()
        c.close()
        return out
    with_arena, without = run(None), run("0")
    for a, b in zip(with_arena, without):
        assert a[:3] == b[:3]
        for x, y in zip(a[3:], b[3:]):
            assert np.array_equal(x, y)
    assert with_arena[0][0] == with_arena[2][0] and np.array_equal(with_arena[0][6], with_arena[2][6])     # first == third solve

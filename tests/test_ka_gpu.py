"""GPU parity for keypoint adjustment: per-edge residuals/Jacobians (pxr_ka_eval) and the
in-kernel bounded LM (pxr_ka_solve) vs the oracle (oracle/pxo_solve.c).
Tolerances: residual/Jacobian 1e-10 relative (north_star: 1e-5); refined keypoints 1e-6 px
while the trajectories coincide (north_star: 1e-4).  Parity unpinned w.r.t. real Ceres."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(ctx, **kw):
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import PatchArena
    from pixsfm_amd.ka_engine import KAProblem
    prob = synthetic_ka.make_ka_problem(**kw)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    return prob, arena, KAProblem(ctx, arena, prob)


def _oracle_edges(prob, cfg_kw=None):
    import pxo
    cfg = pxo.cfg(**(cfg_kw or {}))
    ls = pxo.loss("cauchy", 0.25)
    R, J1s, J2s, costs = [], [], [], []
    for s, d, w in zip(prob["edge_src"], prob["edge_dst"], prob["edge_w"]):
        p1 = pxo.make_patch(prob["patches"][s], prob["corners"][s], prob["scales"][s])
        p2 = pxo.make_patch(prob["patches"][d], prob["corners"][d], prob["scales"][d])
        r, J1, J2 = pxo.ka_residual(p1, p2, cfg, prob["kp"][s], prob["kp"][d])
        R.append(r); J1s.append(J1); J2s.append(J2)
        costs.append(0.5 * pxo.loss_eval(ls, float(r @ r), w)[0])
    return np.array(R), np.array(J1s), np.array(J2s), np.array(costs)


@pytest.mark.parametrize("dtype,channels,scale", [(np.float16, 128, (1.0, 1.0)), (np.float16, 64, (0.5, 0.25)),
                                                   (np.float32, 128, (1.0, 1.0)), (np.float64, 128, (2.0, 1.0))])
def test_edge_residuals_and_jacobians(ctx, dtype, channels, scale):
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=6, track_len=4, seed=11, dtype=dtype, channels=channels, scale=scale)
    cost, r, J1, J2 = ka.eval(interp_cfg(), make_loss("cauchy", [0.25]), materialize=True)
    R, J1o, J2o, co = _oracle_edges(prob)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert rel(r.download(), R) < 1e-10
    assert rel(J1.download(), J1o) < 1e-10
    assert rel(J2.download(), J2o) < 1e-10
    assert rel(cost.download(), co) < 1e-10


def test_solve_matches_oracle_default_options(ctx):
    """KeypointAdjuster defaults (keypoint_adjustment/main.py:61-82): Cauchy(0.25), bound 4,
    parameter_tolerance 1e-5, <= 50 keypoints per sub-problem, roots constant."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=30, track_len=6, seed=5, max_kps_per_problem=50)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
    kp = ka.keypoints()
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0)
    assert len(per) == len(sums) == prob["n_problems"]
    for g, o in zip(per, sums):
        assert g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"]
        assert g["termination"] == o["termination"]
        assert abs(g["initial_cost"] - o["initial_cost"]) < 1e-10 * o["initial_cost"]
        assert abs(g["final_cost"] - o["final_cost"]) < 1e-7 * max(o["final_cost"], 1e-6)
    assert np.abs(kp - kpo).max() < 1e-6
    root = prob["node_const"].astype(bool)
    assert np.array_equal(kp[root], prob["kp"][root])            # roots stay fixed (main.py:175-177)
    assert total["final_cost"] < 0.05 * total["initial_cost"]
    assert abs(total["initial_cost"] - sum(s["initial_cost"] for s in sums)) < 1e-9 * total["initial_cost"]


def test_solve_use_float_simd_and_huber(ctx):
    """InterpolationConfig.use_float_simd (fp32 vertical pass, interpolation.h:620-623) + a non-default loss."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=14, track_len=5, seed=21, max_kps_per_problem=30)
    total, per = ka.solve(interp_cfg(use_float_simd=True), make_loss("huber", [0.3]), bound=4.0, per_problem=True)
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(use_float_simd=True), pxo.loss("huber", 0.3), 4.0)
    for g, o in zip(per, sums):
        assert g["iterations"] == o["iterations"] and g["termination"] == o["termination"]
    assert np.abs(ka.keypoints() - kpo).max() < 1e-6


def test_solve_with_active_bounds_and_many_iterations(ctx):
    """Large detection noise: several nodes end on their +-bound box; tight tolerance so the LM
    runs long.  End points must agree within north_star's 1e-4."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=10, track_len=5, seed=9, sigma=2.5, max_kps_per_problem=25)
    opts = dict(parameter_tolerance=1e-9, max_iterations=40)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=2.0, options=lm_options(**opts),
                          per_problem=True)
    kp = ka.keypoints()
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 2.0, pxo.lm_options(**opts))
    assert np.abs(kp - kpo).max() < 1e-4
    moved = np.abs(kp - prob["kp"]).max(axis=1)
    assert (np.abs(moved - 2.0) < 1e-9).any(), "expected at least one keypoint on its bound"
    assert moved.max() <= 2.0 + 1e-12


def test_oversized_problem_uses_global_matrix(ctx):
    """> 112 unknowns in one sub-problem: the damped matrix falls back from LDS to global scratch."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=16, track_len=5, seed=2, max_kps_per_problem=200)
    assert prob["n_problems"] == 1
    total, _ = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]))
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0)
    assert total["num_point_unknowns"] == sums[0]["num_unknowns"] == 2 * (80 - 16)
    assert np.abs(ka.keypoints() - kpo).max() < 1e-6


def test_infeasible_start_and_empty_problems(ctx):
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    from pixsfm_amd import synthetic_ka
    prob = synthetic_ka.make_ka_problem(n_tracks=4, track_len=3, seed=4, max_kps_per_problem=3)
    # push one free keypoint outside its patch: Ceres declares the problem infeasible and leaves it untouched
    free = np.nonzero(prob["node_const"] == 0)[0][0]
    prob["kp"][free] += 30.0
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ka = KAProblem(ctx, arena, prob)
    total, per = ka.solve(interp_cfg(), make_loss(), per_problem=True)
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss(), 4.0)
    bad = prob["node_problem"][free]
    assert per[bad]["termination"] == 2 == sums[bad]["termination"]
    assert np.abs(ka.keypoints() - kpo).max() < 1e-6
    assert np.array_equal(ka.keypoints()[free], prob["kp"][free])
    # no edges at all
    for k in ("edge_src", "edge_dst", "edge_w"):
        prob[k] = prob[k][:0]
    ka2 = KAProblem(ctx, arena, prob)
    total2, _ = ka2.solve(interp_cfg(), make_loss())
    assert total2["initial_cost"] == 0.0 and np.array_equal(ka2.keypoints(), prob["kp"])


@pytest.mark.parametrize("dtype,channels", [(np.float16, 64), (np.float32, 64), (np.float64, 64), (np.float32, 128),
                                             (np.float64, 128)])
def test_solve_every_storage_type_and_channel_count(ctx, dtype, channels):
    """FeaturePatch storage half / float / double (featurepatch.cc:365-367) x CHANNELS 128 / 64 through the
    in-kernel LM (the fp64 variants run with register spills: parity is what is checked here)."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=8, track_len=4, seed=33, dtype=dtype, channels=channels, max_kps_per_problem=16)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0)
    for g, o in zip(per, sums):
        assert g["iterations"] == o["iterations"] and g["termination"] == o["termination"]
        assert abs(g["final_cost"] - o["final_cost"]) < 1e-7 * max(o["final_cost"], 1e-6)
    assert np.abs(ka.keypoints() - kpo).max() < 1e-6


def test_check_bounds_has_no_effect_on_keypoint_adjustment(ctx):
    """check_bounds with a keypoint outside its patch: the KA functors ignore what PatchInterpolator::Evaluate returns and
    always succeed (featuremetric.h:44-63, feature_reference.h:44-60), so evaluation and solve are those without the option
    -- and the oracle's."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=6, track_len=4, seed=12, max_kps_per_problem=8)
    cost_0 = ka.eval(interp_cfg(), make_loss("cauchy", [0.25]))[0].download()
    bad = int(np.nonzero(prob["node_problem"] == 1)[0][0])
    kp = prob["kp"].copy()
    kp[bad, 0] = prob["corners"][bad, 0] - 3.0                      # left of its patch
    ka.d["kp"].upload(kp)
    c_on = ka.eval(interp_cfg(check_bounds=True), make_loss("cauchy", [0.25]))[0].download()
    c_off = ka.eval(interp_cfg(), make_loss("cauchy", [0.25]))[0].download()
    assert np.isfinite(c_on).all() and np.array_equal(c_on, c_off) and not np.array_equal(c_off, cost_0)
    total_on, per_on = ka.solve(interp_cfg(check_bounds=True), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
    out_on = ka.keypoints()
    ka.d["kp"].upload(kp)
    total_off, per_off = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True)
    assert np.array_equal(out_on, ka.keypoints())
    assert [p["termination"] for p in per_on] == [p["termination"] for p in per_off]
    kpo, sums = pxo_ka.ka_solve(dict(prob, kp=kp), pxo.cfg(check_bounds=True), pxo.loss("cauchy", 0.25), 4.0)
    for g, o in zip(per_on, sums):
        assert g["termination"] == o["termination"] and g["iterations"] == o["iterations"]
    assert np.abs(out_on - kpo).max() < 1e-6

@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
def test_single_channel_features(ctx, dtype):
    """CHANNELS = 1 -- the reference instantiates FeatureMetricKeypointOptimizer for (128, 1) and (1, 1)
    (featuremetric_keypoint_optimizer.h:13-17); below 8 channels it takes the scalar all-fp64 bicubic
    (interpolation.h:222-268).  Scalar fields, l2_normalize off (a 1-vector normalises to +-1): per-edge residuals and
    Jacobians and the bounded LM against the oracle."""
    import pxo
    import pxo_ka
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    prob = synthetic_ka.make_ka_problem(n_tracks=10, track_len=4, seed=21, channels=64, dtype=np.float32, sigma=0.6)
    prob["patches"] = np.ascontiguousarray((3.0 * prob["patches"][..., 5:6]).astype(dtype))     # one smooth channel
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    assert arena.C == 1
    ka = KAProblem(ctx, arena, prob)
    cfg, ls = interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25])
    cost, r, J1, J2 = ka.eval(cfg, ls, materialize=True)
    R, J1o, J2o, co = _oracle_edges(prob, dict(l2_normalize=False))
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert np.abs(R).max() > 1e-3 and np.abs(J1o).max() > 1e-3
    assert rel(r.download(), R) < 1e-12 and rel(J1.download(), J1o) < 1e-12 and rel(J2.download(), J2o) < 1e-12
    assert rel(cost.download(), co) < 1e-12
    total, per = ka.solve(cfg, ls, bound=4.0, per_problem=True)
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(l2_normalize=False), pxo.loss("cauchy", 0.25), 4.0)
    for g, o in zip(per, sums):
        assert g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"]
        assert abs(g["final_cost"] - o["final_cost"]) < 1e-9 * max(o["initial_cost"], 1e-12)
    assert np.abs(ka.keypoints() - kpo).max() < 1e-6
    assert total["final_cost"] < total["initial_cost"]
    arena.close()


def test_sub_problem_larger_than_the_lds_metadata_caches(ctx):
    """One sub-problem of 270 nodes / 1350 residual blocks: the solve kernel keeps the metadata of the first 96 nodes and 512
    blocks in LDS and reads the rest from global memory -- both paths in one solve, against the oracle."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob, arena, ka = _setup(ctx, n_tracks=45, track_len=6, seed=77, max_kps_per_problem=100000)
    assert prob["n_problems"] == 1 and len(prob["kp"]) > 96 and len(prob["edge_src"]) > 512
    opts = dict(max_iterations=6, parameter_tolerance=1e-5)
    total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, per_problem=True, options=lm_options(**opts))
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0, pxo.lm_options(**opts))
    assert per[0]["iterations"] == sums[0]["iterations"] and per[0]["num_successful"] == sums[0]["num_successful"]
    assert abs(per[0]["final_cost"] - sums[0]["final_cost"]) < 1e-8 * sums[0]["final_cost"]
    assert np.abs(ka.keypoints() - kpo).max() < 1e-6


def test_all_constant_sub_problem_reports_its_cost(ctx):
    """Every keypoint of one sub-problem held constant (n = 0 unknowns): nothing to solve, but the summary's initial / final
    cost is the cost of its residual blocks -- evaluated on ALL nodes (the value-only pass of the line-search probes skips
    constant nodes; this path must not)."""
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    from pixsfm_amd import synthetic_ka
    prob = synthetic_ka.make_ka_problem(n_tracks=6, track_len=4, seed=9, max_kps_per_problem=8)
    frozen = int(prob["node_problem"].max())
    prob["node_const"] = np.where(prob["node_problem"] == frozen, 1, prob["node_const"]).astype(np.uint8)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ka = KAProblem(ctx, arena, prob)
    total, per = ka.solve(interp_cfg(), make_loss(), per_problem=True)
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss(), 4.0)
    assert per[frozen]["num_camera_unknowns"] == 0 and per[frozen]["initial_cost"] > 0
    assert abs(per[frozen]["initial_cost"] - sums[frozen]["initial_cost"]) < 1e-10 * sums[frozen]["initial_cost"]
    assert per[frozen]["final_cost"] == per[frozen]["initial_cost"]
    assert np.array_equal(ka.keypoints()[prob["node_problem"] == frozen], prob["kp"][prob["node_problem"] == frozen])
    assert np.abs(ka.keypoints() - kpo).max() < 1e-6


# ---- label groups that span several workgroups (pxr_ka_view.d_prob_group, round 6) ----------------------------------------------
def _grouped(ctx, prob):
    """the same problem twice: its label groups on ONE workgroup each, and chunked by tracks (node_track given)"""
    from pixsfm_amd.engine import PatchArena
    from pixsfm_amd.ka_engine import KAProblem
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    single = KAProblem(ctx, arena, prob)
    chunked = KAProblem(ctx, arena, dict(prob, node_track=prob["track_of_node"]))
    return arena, single, chunked


@pytest.mark.parametrize("bound,sigma", [(4.0, 1.0), (1.5, 1.5)])
def test_chunked_label_group_takes_the_decisions_of_one_problem(ctx, bound, sigma):
    """One label group of 300 keypoints (60 tracks): as ONE sub-problem on one workgroup, and as 8 chunks of whole tracks on 8
    workgroups that share the trust region.  Same iterations, accepted steps, termination and costs as the oracle's solve of
    the one problem (one Ceres problem in the reference: keypoint_adjustment/main.py:197-202 with split_in_subproblems = false,
    or a 1000-keypoint group of configs/low_memory.yaml); keypoints to 1e-6 px.  sigma 1.5 / bound 1.5: active bounds, twenty-probe
    line searches -- every probe's cost is a sum over the chunks."""
    import pxo
    import pxo_ka
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import interp_cfg, lm_options, make_loss
    prob = synthetic_ka.make_ka_problem(n_tracks=60, track_len=5, seed=21, max_kps_per_problem=100000, sigma=sigma)
    assert prob["n_problems"] == 1
    arena, single, chunked = _grouped(ctx, prob)
    assert chunked.problem_group is not None and len(chunked.problem_group) == 6 and single.problem_group is None
    opts = lm_options(parameter_tolerance=1e-5)
    t1, p1 = single.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=bound, options=opts, per_problem=True)
    t2, p2 = chunked.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=bound, options=opts, per_problem=True)
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), bound, pxo.lm_options(parameter_tolerance=1e-5))
    assert len(p2) == 1 == len(sums)
    for got in (p1[0], p2[0]):
        assert got["iterations"] == sums[0]["iterations"] and got["num_successful"] == sums[0]["num_successful"]
        assert got["termination"] == sums[0]["termination"]
        assert abs(got["initial_cost"] - sums[0]["initial_cost"]) < 1e-10 * sums[0]["initial_cost"]
        assert abs(got["final_cost"] - sums[0]["final_cost"]) < 1e-7 * max(sums[0]["final_cost"], 1e-6)
    assert t2["iterations"] == t1["iterations"] and t2["num_successful"] == t1["num_successful"]
    assert np.abs(chunked.keypoints() - kpo).max() < 1e-6 and np.abs(single.keypoints() - kpo).max() < 1e-6
    assert np.abs(chunked.keypoints() - single.keypoints()).max() < 1e-8
    arena.close()


def test_chunked_groups_of_different_sizes_next_to_small_ones_and_twice_the_same_bits(ctx):
    """Three label groups: 200 keypoints (4 chunks), 40 keypoints (stays one sub-problem), 130 keypoints (3 chunks).  Every group
    ends like the oracle's solve of that group as one problem; the deterministic default gives the same bits on a second run."""
    import pxo
    import pxo_ka
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    prob = synthetic_ka.make_ka_problem(n_tracks=74, track_len=5, seed=33, max_kps_per_problem=100000)
    track = prob["track_of_node"]
    prob["node_problem"] = np.where(track < 40, 0, np.where(track < 48, 1, 2)).astype(np.int32)
    prob["n_problems"] = 3
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    out = []
    for _ in range(2):
        ka = KAProblem(ctx, arena, dict(prob, node_track=track))
        assert ka.problem_group.tolist() == [0, 0, 0, 0, 1, 2, 2, 2]
        total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=4.0, options=lm_options(parameter_tolerance=1e-5), per_problem=True)
        out.append((ka.keypoints(), total, per))
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), 4.0, pxo.lm_options(parameter_tolerance=1e-5))
    kp, total, per = out[0]
    assert len(per) == 3
    for g, o in zip(per, sums):
        assert g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"] and g["termination"] == o["termination"]
        assert abs(g["final_cost"] - o["final_cost"]) < 1e-7 * max(o["final_cost"], 1e-6)
    assert np.abs(kp - kpo).max() < 1e-6
    assert total["num_successful"] == sum(o["num_successful"] for o in sums)
    assert ctx.deterministic and np.array_equal(out[0][0], out[1][0]) and out[0][1]["final_cost"] == out[1][1]["final_cost"]
    arena.close()


def test_a_group_with_too_many_chunks_is_refused_through_the_c_abi(ctx):
    """More chunks in one group than a launch keeps resident: the spin-waits of the group sums could deadlock, so the C-ABI
    refuses (the Python layer never builds such a view: chunk_label_groups leaves such a group on one workgroup)."""
    from pixsfm_amd import PixsfmHipError, synthetic_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    prob = synthetic_ka.make_ka_problem(n_tracks=600, track_len=2, seed=3, max_kps_per_problem=2, channels=64)
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ka = KAProblem(ctx, arena, prob)
    assert ka.n_problems == 600
    ka.d["prob_group"] = ctx.to_device(np.zeros(600, np.int32), np.int32)
    ka.view.d_prob_group = ka.d["prob_group"].ptr
    with pytest.raises(PixsfmHipError, match="resident"):
        ka.solve(interp_cfg(), make_loss("cauchy", [0.25]))
    arena.close()


# ---- the two-phase launch (ka_solve_kernel_sched, round 6) ------------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["persistent", "two_launches_1", "two_launches_2"])
@pytest.mark.parametrize("sigma,bound", [(1.0, 4.0), (1.5, 1.5)])
def test_two_phase_launch_gives_the_one_phase_results_bit_for_bit(ctx, monkeypatch, sigma, bound, variant):
    """More sub-problems than resident workgroups (forced here: a grid of 5): every sub-problem runs ONE LM iteration, parks its LM
    state, and is resumed from a list that starts with those sitting on a bound.  Same arithmetic per sub-problem -- the resumed
    linearisation is accumulated on the grid the interrupted one used -- so keypoints, costs and counts equal the one-phase
    launch's bit for bit (deterministic default), and the oracle's to the usual tolerances."""
    import pxo
    import pxo_ka
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    prob = synthetic_ka.make_ka_problem(n_tracks=90, track_len=5, seed=17, max_kps_per_problem=15, sigma=sigma)
    assert prob["n_problems"] == 30
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    out = []
    for knob in ("0", "1"):
        # (the persistent grid, or two launches of the plain kernel that park after 1 / 2 LM iterations: PXR_KA_TWO_LAUNCH)
        monkeypatch.setenv("PXR_KA_TWO_PHASE", knob if variant == "persistent" else "0")
        monkeypatch.setenv("PXR_KA_TWO_LAUNCH", "0" if (variant == "persistent" or knob == "0") else variant[-1])
        monkeypatch.setenv("PXR_KA_TWO_PHASE_RESIDENT", "5")
        ka = KAProblem(ctx, arena, prob)
        total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=bound, options=lm_options(parameter_tolerance=1e-5), per_problem=True)
        out.append((ka.keypoints(), total, per))
    (kp0, t0, per0), (kp1, t1, per1) = out
    assert ctx.deterministic and np.array_equal(kp0, kp1)
    for a, b in zip(per0, per1):
        for k in ("iterations", "num_successful", "termination", "initial_cost", "final_cost"):
            assert a[k] == b[k], (k, a[k], b[k])
        assert b["linear_iterations"] >= a["linear_iterations"]        # (a resumed sub-problem interpolates its nodes once more)
    assert t0["final_cost"] == t1["final_cost"] and t0["num_successful"] == t1["num_successful"]
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), pxo.loss("cauchy", 0.25), bound, pxo.lm_options(parameter_tolerance=1e-5))
    for g, o in zip(per1, sums):
        assert g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"] and g["termination"] == o["termination"]
    assert np.abs(kp1 - kpo).max() < 1e-6
    arena.close()


def test_two_phase_launch_with_floating_point_atomics(monkeypatch):
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import Context, PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    c = Context(0)
    c.deterministic = False
    prob = synthetic_ka.make_ka_problem(n_tracks=60, track_len=5, seed=18, max_kps_per_problem=10, sigma=1.5)
    arena = PatchArena.from_numpy(c, prob["patches"], prob["corners"], prob["scales"])
    out = []
    for knob in ("0", "1", "launches"):
        monkeypatch.setenv("PXR_KA_TWO_PHASE", "1" if knob == "1" else "0")
        monkeypatch.setenv("PXR_KA_TWO_LAUNCH", "2" if knob == "launches" else "0")
        monkeypatch.setenv("PXR_KA_TWO_PHASE_RESIDENT", "7")
        ka = KAProblem(c, arena, prob)
        total, per = ka.solve(interp_cfg(), make_loss("cauchy", [0.25]), bound=2.0, options=lm_options(parameter_tolerance=1e-5), per_problem=True)
        out.append((ka.keypoints(), total, per))
    for other in out[1:]:
        assert np.abs(out[0][0] - other[0]).max() < 1e-9
        assert [p["iterations"] for p in out[0][2]] == [p["iterations"] for p in other[2]]
        assert abs(out[0][1]["final_cost"] - other[1]["final_cost"]) < 1e-10 * out[0][1]["initial_cost"]
    arena.close(); c.close()


@pytest.mark.parametrize("scale", [300.0, 3000.0])
def test_two_launches_with_a_fixed_point_rescale_in_between(ctx, monkeypatch, scale):
    """Raw features 300x / 3000x unit norm: sub-problems stop with the internal 'coarser grid' code in EITHER launch of the two-launch
    schedule and the solve is launched again.  Parked state and rescale state share one record per sub-problem: the result must
    still be the one-launch result bit for bit."""
    from pixsfm_amd import synthetic_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    prob = dict(synthetic_ka.make_ka_problem(n_tracks=90, track_len=5, seed=23, max_kps_per_problem=15, sigma=1.5))
    prob["patches"] = (prob["patches"].astype(np.float32) * scale).astype(np.float16)
    assert np.isfinite(prob["patches"].astype(np.float32)).all()
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    out = []
    for knob in ("0", "1", "2", "3"):
        monkeypatch.setenv("PXR_KA_TWO_LAUNCH", knob)
        monkeypatch.setenv("PXR_KA_TWO_PHASE_RESIDENT", "5")
        ka = KAProblem(ctx, arena, prob)
        total, per = ka.solve(interp_cfg(l2_normalize=False), make_loss("cauchy", [0.25 * scale]), bound=3.0,
                              options=lm_options(parameter_tolerance=1e-5), per_problem=True)
        out.append((ka.keypoints(), total, per))
    assert ctx.deterministic
    for kp, total, per in out[1:]:
        assert np.array_equal(out[0][0], kp)
        assert total["final_cost"] == out[0][1]["final_cost"] and total["num_successful"] == out[0][1]["num_successful"]
        for a, b in zip(out[0][2], per):
            for k in ("iterations", "num_successful", "termination", "initial_cost", "final_cost"):
                assert a[k] == b[k], (k, a[k], b[k])
    assert out[0][1]["final_cost"] < out[0][1]["initial_cost"]
    arena.close()

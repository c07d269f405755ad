"""GPU parity for the block-diagonal KA solver: unary reference terms (localization QKA,
FeatureReference2DCostFunctor), mixed unary + pairwise problems, and component detection on
graphs that are not complete (chains, components beyond one wavefront) -- all vs the oracle.
Tolerance: refined keypoints 1e-6 px while the trajectories coincide (north_star: 1e-4)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _descriptors_at(prob, nodes, xy, cfg):
    """f(patch[node], xy) through the oracle's FeatureReference2D functor with a zero reference."""
    import pxo
    out = []
    zero = np.zeros(prob["patches"].shape[-1])
    for nd, p in zip(nodes, xy):
        patch = pxo.make_patch(prob["patches"][nd], prob["corners"][nd], prob["scales"][nd])
        r, _ = pxo.ref2d_residual(patch, cfg, p, zero)
        out.append(r)
    return np.array(out)


def _qka_problem(n_kp, seed, stacked=False):
    """One query image: n_kp keypoints, each with the descriptor of its 3D point's reference (taken
    from the other observation of the same synthetic track at its true location)."""
    import pxo
    from pixsfm_amd import synthetic_ka
    base = synthetic_ka.make_ka_problem(n_tracks=n_kp, track_len=2, seed=seed, sigma=0.8)
    q = np.arange(0, 2 * n_kp, 2)
    refs = _descriptors_at(base, q + 1, base["true_xy"][q + 1], pxo.cfg())
    prob = dict(base)
    prob.update(kp=base["kp"][q].copy(), node_patch=q.astype(np.int64), node_const=np.zeros(n_kp, np.uint8),
                node_problem=np.zeros(n_kp, np.int32), edge_src=np.zeros(0, np.int32), edge_dst=np.zeros(0, np.int32),
                edge_w=np.zeros(0), unary_node=np.arange(n_kp, dtype=np.int32), unary_ref=refs, unary_w=None,
                true_xy=base["true_xy"][q], n_problems=1)
    if stacked:   # a second reference on every third keypoint (refine_stacked, localization/main.py:158-192)
        extra = np.arange(0, n_kp, 3, dtype=np.int32)
        rng = np.random.default_rng(seed)
        refs2 = refs[extra] + rng.normal(0, 0.01, refs[extra].shape)
        prob["unary_node"] = np.concatenate([prob["unary_node"], extra])
        prob["unary_ref"] = np.concatenate([refs, refs2])
    return prob


def _solve_both(ctx, prob, loss=("trivial", []), bound=4.0, opts_kw=None):
    import pxo
    import pxo_ka
    from pixsfm_amd.engine import PatchArena, interp_cfg, lm_options, make_loss
    from pixsfm_amd.ka_engine import KAProblem
    arena = PatchArena.from_numpy(ctx, prob["patches"], prob["corners"], prob["scales"])
    ka = KAProblem(ctx, arena, prob)
    kw = dict(parameter_tolerance=1e-5)
    kw.update(opts_kw or {})
    total, per = ka.solve(interp_cfg(), make_loss(*loss), bound=bound, options=lm_options(**kw), per_problem=True)
    ols = pxo.loss(loss[0], *(loss[1] or [])) if loss[0] != "trivial" else pxo.loss("trivial")
    kpo, sums = pxo_ka.ka_solve(prob, pxo.cfg(), ols, bound, pxo.lm_options(**kw))
    return ka.keypoints(), per, kpo, sums, total


def _check(per, sums, kp, kpo, tol=1e-6):
    assert len(per) == len(sums)
    for g, o in zip(per, sums):
        assert g["iterations"] == o["iterations"] and g["num_successful"] == o["num_successful"]
        assert g["termination"] == o["termination"]
        assert g["num_camera_unknowns"] == o["num_unknowns"]
        assert abs(g["initial_cost"] - o["initial_cost"]) <= 1e-10 * max(o["initial_cost"], 1e-12)
        assert abs(g["final_cost"] - o["final_cost"]) <= 1e-7 * max(o["final_cost"], 1e-6)
    assert np.abs(kp - kpo).max() < tol


@pytest.mark.parametrize("n_kp,stacked", [(7, False), (300, False), (40, True)])
def test_query_keypoint_adjustment_matches_oracle(ctx, n_kp, stacked):
    """QueryKeypointAdjuster defaults (localization/main.py:89-108): trivial loss, bound 4,
    parameter_tolerance 1e-5; one problem over all keypoints of the query."""
    prob = _qka_problem(n_kp, seed=3 + n_kp, stacked=stacked)
    kp, per, kpo, sums, total = _solve_both(ctx, prob)
    _check(per, sums, kp, kpo)
    err0 = np.linalg.norm(prob["kp"] - prob["true_xy"], axis=1)
    err1 = np.linalg.norm(kp - prob["true_xy"], axis=1)
    assert np.median(err1) < 0.1 * np.median(err0)
    assert total["num_point_unknowns"] == 2 * n_kp


def test_mixed_unary_and_pairwise_terms(ctx):
    """Pairwise KA edges plus weighted unary references on a subset of the nodes, Cauchy loss."""
    import pxo
    from pixsfm_amd import synthetic_ka
    prob = synthetic_ka.make_ka_problem(n_tracks=12, track_len=5, seed=8, max_kps_per_problem=25)
    rng = np.random.default_rng(1)
    nodes = rng.choice(len(prob["kp"]), 20, replace=False).astype(np.int32)
    nodes = np.concatenate([nodes, nodes[:5]])                       # stacked terms
    refs = _descriptors_at(prob, nodes, prob["true_xy"][nodes], pxo.cfg())
    prob.update(unary_node=nodes, unary_ref=refs, unary_w=rng.uniform(0.5, 2.0, len(nodes)))
    kp, per, kpo, sums, _ = _solve_both(ctx, prob, loss=("cauchy", [0.25]))
    _check(per, sums, kp, kpo)
    # a constant root with a unary term contributes cost but no unknowns
    assert prob["node_const"][nodes].any()


@pytest.mark.parametrize("track_len,chain", [(9, True), (45, True), (40, False)])
def test_components_on_chains_and_beyond_one_wavefront(ctx, track_len, chain):
    """Chain graphs need several rounds of label propagation; 45-node chains / 40-node complete
    tracks give components of 88 / 78 unknowns (> 64: the workgroup-wide factorisation)."""
    from pixsfm_amd import synthetic_ka
    prob = synthetic_ka.make_ka_problem(n_tracks=3, track_len=track_len, seed=17, max_kps_per_problem=-1,
                                        directed_both=not chain)
    if chain:
        src = np.concatenate([np.arange(t * track_len, (t + 1) * track_len - 1) for t in range(3)]).astype(np.int32)
        rng = np.random.default_rng(2)
        rng.shuffle(src)
        prob.update(edge_src=src + 1, edge_dst=src, edge_w=rng.uniform(0.5, 1.0, len(src)))
    prob["node_problem"] = np.zeros(len(prob["kp"]), np.int32)       # all three tracks in one sub-problem
    prob["n_problems"] = 1
    kp, per, kpo, sums, _ = _solve_both(ctx, prob, loss=("cauchy", [0.25]), opts_kw=dict(max_iterations=30))
    _check(per, sums, kp, kpo, tol=1e-5)
    assert per[0]["num_camera_unknowns"] == 2 * (3 * track_len - 3)

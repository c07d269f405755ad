"""Keypoint-adjustment problem construction (the RunSubset semantics of A15) on 25 seeded match graphs
under varying options (tests/cases/ka_setup_cases.py): the product's native edge builder (pxr_ka_build_edges), its node-role
logic (api.keypoint_adjustment.node_roles) and the oracle's C bounds (oracle/pxo_solve.c ka_bounds, which the GPU kernel is
compared with in tests/test_ka_gpu.py) against the oracle's Python restatement of what the reference hands to Ceres
(oracle/pxo_graph.py: TopologicalKeypointOptimizer::SetUp, FeatureMetricKeypointOptimizer::AddIntraResiduals,
KeypointOptimizerBase::ParameterizeKeypoints, KeypointAdjustmentSetup): residual blocks with their ScaledLoss weights (as a
multiset: the reference walks an unordered_set), constant keypoints, box bounds, touched-but-unvisited free blocks.
PARITY UNPINNED: the reference has no test for this construction and its headers cannot be compiled here."""
import numpy as np

from cases import ka_setup_cases as gen_mod


def _product_graph(c):
    from pixsfm_amd.api import base
    g = base.Graph()
    for (a, b), (matches, sims) in zip(c["pairs"], c["mm"]):
        g.register_matches("im%d" % a, "im%d" % b, matches, sims)
    return g


def oracle_setup(c):
    """{src, dst, w (canonical order), const, bounds} of the oracle's restatement for one case."""
    import pxo_graph
    g = pxo_graph.Graph()
    for (a, b), (matches, sims) in zip(c["pairs"], c["mm"]):
        g.register_matches("im%d" % a, "im%d" % b, matches, sims)
    sub = None if c["nodes_in_problem"] is None else [int(i) for i in c["nodes_in_problem"]]
    src, dst, w = pxo_graph.build_edges(g, None, c["labels"], c["roots"], sub, c["weight_by_sim"], c["root_edges_only"],
                                        c["root_regularize_weight"])
    const_images = {g.image_name_to_id["im%d" % k] for k in c["const_images"] if "im%d" % k in g.image_name_to_id}

    def is_const(node):                       # KeypointAdjustmentSetup::IsNodeConstant: constant image, or a masked (root) node
        return node.image_id in const_images or (c["const_roots"] and bool(c["roots"][node.node_idx]))
    node_kp = c["kp"][c["kp_ptr"][c["node_image"]] + c["node_feature"]]
    const, bounds = pxo_graph.parameterize_keypoints(g, src, dst, sub, is_const, node_kp, c["corner"], c["scale"], 16, 16, c["bound"])
    src, dst, w = np.asarray(src, np.int64), np.asarray(dst, np.int64), np.asarray(w, np.float64)
    order = np.lexsort((w, dst, src))
    return dict(src=src[order], dst=dst[order], w=w[order], const=const, bounds=bounds)


def _check_case(c, want):
    import pxo_ka
    from pixsfm_amd.api.keypoint_adjustment import KeypointAdjustmentSetup, build_edges, node_roles
    g = _product_graph(c)
    n = len(g.nodes)
    assert [nd.feature_idx for nd in g.nodes] == c["node_feature"].tolist()
    keypoints = {"im%d" % k: c["kp"][c["kp_ptr"][k]:c["kp_ptr"][k + 1]].copy() for k in range(c["n_images"])}
    sub = None if c["nodes_in_problem"] is None else [int(i) for i in c["nodes_in_problem"]]
    src, dst, w = build_edges(g, keypoints, c["labels"], c["roots"], sub, c["weight_by_sim"], c["root_edges_only"],
                              c["root_regularize_weight"])
    order = np.lexsort((w, dst, src))
    src, dst, w = np.asarray(src, dtype=np.int64)[order], np.asarray(dst, dtype=np.int64)[order], np.asarray(w, dtype=np.float64)[order]
    assert np.array_equal(src, want["src"]) and np.array_equal(dst, want["dst"]), c["name"]
    assert np.array_equal(w, want["w"]), c["name"]
    # constant / boxed / free keypoints
    setup = KeypointAdjustmentSetup()
    for k in c["const_images"]:
        if "im%d" % k in g.image_name_to_id:
            setup.set_image_constant(g.image_name_to_id["im%d" % k])
    if c["const_roots"]:
        setup.set_masked_nodes_constant(g, [bool(r) for r in c["roots"]])
    roles, in_solve = node_roles(setup, g, src, dst, sub)
    touched = np.zeros(n, bool)
    touched[src] = True; touched[dst] = True
    has_bounds = ~np.isnan(want["bounds"]).any(1)
    node_kp = c["kp"][c["kp_ptr"][c["node_image"]] + c["node_feature"]]
    bounds = pxo_ka.node_bounds(node_kp, c["corner"], c["scale"], 16, 16, c["bound"])
    n_checked = 0
    for i in np.flatnonzero(touched):
        assert in_solve[i]
        if want["const"][i]:
            assert roles[i] == 1, (c["name"], i)
        elif has_bounds[i]:
            assert roles[i] == 0, (c["name"], i)
            assert np.array_equal(bounds[i], want["bounds"][i]), (c["name"], i, bounds[i], want["bounds"][i])
        else:
            assert roles[i] == 2, (c["name"], i)          # a parameter block ParameterizeKeypoints never visited
        n_checked += 1
    # nodes outside every residual block are no parameter blocks of the reference's problem
    assert not want["const"][~touched].any() and not has_bounds[~touched].any()
    return n_checked, int((roles[touched] == 2).sum()), int(want["const"].sum())


def test_edges_roles_and_bounds_match_the_oracles_restatement():
    tot = free = const = blocks = 0
    for c in gen_mod.cases():
        want = oracle_setup(c)
        a, b, d = _check_case(c, want)
        tot += a; free += b; const += d; blocks += len(want["src"])
    assert blocks > 1000 and tot > 500 and free > 0 and const > 50

"""Keypoint-adjustment problem construction (the RunSubset semantics of A15) pinned against the
REFERENCE's own code: tests/golden/ka_setup_ref.npz holds what pixsfm's TopologicalKeypointOptimizer::SetUp,
FeatureMetricKeypointOptimizer::AddIntraResiduals and KeypointOptimizerBase::ParameterizeKeypoints hand to Ceres -- recorded by
a stub ceres::Problem (tests/golden/make_golden_ka_setup.py, oracle/ref_ka_setup_shim.cc) -- for 25 seeded match graphs under
varying options: residual blocks with their ScaledLoss weights, constant keypoints, box bounds.
Checked here: the product's native edge builder (pxr_ka_build_edges), its node-role logic (api.keypoint_adjustment.node_roles)
and the oracle's bounds (oracle/pxo_solve.c ka_bounds, which the GPU kernel is compared with in tests/test_ka_gpu.py)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_ka_setup", os.path.join(HERE, "golden", "make_golden_ka_setup.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _product_graph(c):
    from pixsfm_amd.api import base
    g = base.Graph()
    for (a, b), (matches, sims) in zip(c["pairs"], c["mm"]):
        g.register_matches("im%d" % a, "im%d" % b, matches, sims)
    return g


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
    # the residual blocks and their ScaledLoss weights (as a multiset: the reference walks an unordered_set)
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


def test_edges_roles_and_bounds_match_the_reference_vectors():
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "ka_setup_ref.npz"))
    tot = free = const = blocks = 0
    for c in gen.cases():
        want = {k: gold[c["name"] + "_" + k] for k in ("src", "dst", "w", "const", "bounds")}
        a, b, d = _check_case(c, want)
        tot += a; free += b; const += d; blocks += len(want["src"])
    assert blocks > 1000 and tot > 500 and free > 0 and const > 50


def test_reference_run_live_when_present():
    gen = _gen()
    if not os.path.exists(gen.LIB):
        pytest.skip("oracle/_ref/libpxo_ref_ka_setup.so not built (reference tree absent)")
    gold = np.load(os.path.join(HERE, "golden", "ka_setup_ref.npz"))
    for c in gen.cases()[::4]:
        r = gen.run_reference(c)
        for k, v in r.items():
            assert np.array_equal(v, gold[c["name"] + "_" + k], equal_nan=True), (c["name"], k)

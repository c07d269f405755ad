"""Configuration defaults of the pixsfm-shaped API against the REFERENCE's own: tests/golden/default_conf_ref.json holds the
`default_conf` dictionaries of the reference's adjuster classes and the base interpolation / solver defaults, read off the
classes after importing the reference's unmodified main.py modules (tests/golden/make_golden_defaults.py).  A drop-in must
start from the same numbers: tolerances, bounds, sub-problem sizes, loss scales, refine_* flags, ..."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _diff(a, b, path=""):
    out = []
    if isinstance(a, dict) and isinstance(b, dict):
        for k in sorted(set(a) | set(b)):
            if k not in a:
                out.append("%s/%s: only in the product = %r" % (path, k, b[k]))
            elif k not in b:
                out.append("%s/%s: only in the reference = %r" % (path, k, a[k]))
            else:
                out += _diff(a[k], b[k], path + "/" + k)
    elif json.loads(json.dumps(a, default=str)) != json.loads(json.dumps(b, default=str)):
        out.append("%s: reference %r, product %r" % (path, a, b))
    return out


def test_default_configurations_equal_the_reference():
    from pixsfm_amd.api import base, bundle_adjustment as ba, keypoint_adjustment as ka
    ref = json.load(open(os.path.join(HERE, "golden", "default_conf_ref.json")))
    mine = {"KeypointAdjuster": ka.KeypointAdjuster.default_conf,
            "FeatureMetricKeypointAdjuster": ka.FeatureMetricKeypointAdjuster.default_conf,
            "TopologicalReferenceKeypointAdjuster": ka.TopologicalReferenceKeypointAdjuster.default_conf,
            "BundleAdjuster": ba.BundleAdjuster.default_conf,
            "FeatureReferenceBundleAdjuster": ba.FeatureReferenceBundleAdjuster.default_conf,
            "CostMapBundleAdjuster": ba.CostMapBundleAdjuster.default_conf,
            "interpolation_default_conf": base.interpolation_default_conf, "solver_default_conf": base.solver_default_conf}
    assert sorted(ref) == sorted(mine)
    problems = [line for k in ref for line in _diff(ref[k], mine[k], k)]
    assert not problems, "\n".join(problems)


def test_reference_defaults_live_when_present():
    import importlib.util
    import pytest
    if not os.path.isdir("/root/reference/pixsfm"):
        pytest.skip("reference tree absent")
    spec = importlib.util.spec_from_file_location("make_golden_defaults", os.path.join(HERE, "golden", "make_golden_defaults.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    live = json.loads(json.dumps(m.collect(), default=str, sort_keys=True))
    assert live == json.load(open(os.path.join(HERE, "golden", "default_conf_ref.json")))


def test_option_structs_default_like_the_cxx_structs():
    """The optimizer / extractor classes stand for the reference's C++ option structs (constructed from a possibly PARTIAL dict
    by make_dataclass): a key the caller leaves out takes the STRUCT's default, which differs from the Python-level default_conf
    of the adjusters in a few places.  Values read off the struct definitions."""
    from pixsfm_amd.api import bundle_adjustment as ba, keypoint_adjustment as ka, localization as loc
    # KeypointOptimizerOptions (keypoint_adjustment_options.h:46-80) + FeatureMetricKeypointOptimizer::Options (featuremetric_keypoint_optimizer.h:30-36)
    o = ka.FeatureMetricKeypointOptimizer.option_defaults
    assert o["loss"] == {"name": "cauchy", "params": [0.25]} and o["bound"] == -1.0 and o["print_summary"] is True and o["num_threads"] == -1
    assert (o["solver"]["max_num_iterations"], o["solver"]["max_num_consecutive_invalid_steps"], o["solver"]["function_tolerance"],
            o["solver"]["gradient_tolerance"], o["solver"]["parameter_tolerance"], o["solver"]["num_threads"]) == (100, 10, 0.0, 0.0, 1e-4, 1)
    assert (o["root_regularize_weight"], o["weight_by_sim"], o["root_edges_only"]) == (-1.0, True, False)
    # TopologicalReferenceKeypointOptimizer::Options (topological_reference_keypoint_optimizer.h:8-15) overrides three of them
    t = ka.TopologicalReferenceKeypointOptimizer({}, None, None).options
    assert (t["weight_by_sim"], t["root_regularize_weight"], t["root_edges_only"]) == (False, 1.0, True)
    # BundleOptimizerOptions (bundle_adjustment_options.h:44-96)
    o = ba.FeatureReferenceBundleOptimizer.option_defaults
    assert o["loss"] == {"name": "cauchy", "params": [0.25]} and o["print_summary"] is True and o["min_track_length"] == -1
    assert (o["refine_focal_length"], o["refine_principal_point"], o["refine_extra_params"], o["refine_extrinsics"]) == (True, False, True, True)
    assert (o["solver"]["function_tolerance"], o["solver"]["gradient_tolerance"], o["solver"]["parameter_tolerance"], o["solver"]["max_num_iterations"],
            o["solver"]["max_linear_solver_iterations"], o["solver"]["max_num_consecutive_invalid_steps"],
            o["solver"]["max_consecutive_nonmonotonic_steps"]) == (0.0, 0.0, 0.0, 100, 200, 10, 10)
    # ReferenceConfig (reference_extractor.h:55-67), CostMapConfig (costmap_extractor.h:38-62)
    r = ba.ReferenceExtractor.default_conf
    assert r["loss"] == {"name": "cauchy", "params": [0.25]} and (r["iters"], r["keep_observations"], r["compute_offsets3D"], r["num_threads"]) == (10, False, False, -1)
    c = ba.CostMapExtractor.default_conf
    assert c["loss"] == {"name": "trivial", "params": []} and (c["upsampling_factor"], c["as_gradientfield"], c["compute_cross_derivative"],
                                                               c["apply_sqrt"], c["num_threads"], c["dense_cut_size"]) == (1.0, True, False, False, -1, 12)
    # QueryKeypointOptimizerOptions / QueryBundleOptimizerOptions (query_refinement_options.h:60-95, :8-57)
    q = loc.QueryKeypointOptimizer.option_defaults
    assert q["loss"] == {"name": "trivial", "params": []} and q["bound"] == -1.0 and q["print_summary"] is True
    assert (q["solver"]["max_num_iterations"], q["solver"]["max_num_consecutive_invalid_steps"], q["solver"]["parameter_tolerance"]) == (100, 10, 1e-4)
    b = loc.QueryBundleOptimizer.option_defaults
    assert b["loss"] == {"name": "cauchy", "params": [0.25]} and b["print_summary"] is True
    assert (b["refine_focal_length"], b["refine_principal_point"], b["refine_extra_params"]) == (False, False, False)
    assert (b["solver"]["parameter_tolerance"], b["solver"]["max_num_iterations"], b["solver"]["max_linear_solver_iterations"]) == (1e-5, 100, 200)

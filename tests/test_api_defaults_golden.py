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

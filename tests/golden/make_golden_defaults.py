"""Generates tests/golden/default_conf_ref.json: the `default_conf` dictionaries of the REFERENCE's own adjuster classes
(pixsfm/keypoint_adjustment/main.py, pixsfm/bundle_adjustment/main.py, pixsfm/base/main.py), read from the classes after
importing those unmodified modules from /root/reference on top of the `_pixsfm` adapter and the stand-ins of
tests/test_pixsfm_shim.py (omegaconf / pyceres / pycolmap are absent here).

Run in the build container only:  python tests/golden/make_golden_defaults.py
"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pixel-perfect-sfm_amd"))


def collect():
    import test_pixsfm_shim as shim_test
    gen = shim_test.reference_pixsfm.__wrapped__()          # the fixture's generator
    imp = next(gen)
    out = {}
    ka = imp("pixsfm.keypoint_adjustment.main")
    ba = imp("pixsfm.bundle_adjustment.main")
    base = imp("pixsfm.base")
    unwrap = shim_test._Cfg.unwrap
    for mod, names in ((ka, ("KeypointAdjuster", "FeatureMetricKeypointAdjuster", "TopologicalReferenceKeypointAdjuster")),
                       (ba, ("BundleAdjuster", "FeatureReferenceBundleAdjuster", "CostMapBundleAdjuster"))):
        for n in names:
            cls = getattr(mod, n, None)
            if cls is not None and hasattr(cls, "default_conf"):
                out[n] = unwrap(cls.default_conf)
    for n in ("interpolation_default_conf", "solver_default_conf"):
        if hasattr(base, n):
            out[n] = unwrap(getattr(base, n))
    try:
        next(gen)
    except StopIteration:
        pass
    return out


if __name__ == "__main__":
    conf = collect()
    path = os.path.join(HERE, "default_conf_ref.json")
    with open(path, "w") as f:
        json.dump(conf, f, indent=1, sort_keys=True, default=str)
    print("wrote", path, sorted(conf))

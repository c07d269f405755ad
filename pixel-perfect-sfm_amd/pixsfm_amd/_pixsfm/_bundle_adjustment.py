"""`_pixsfm._bundle_adjustment` (pixsfm/bundle_adjustment/bindings.cc:20-184)."""
from ..api.bundle_adjustment import (BundleAdjustmentSetup, CostMapBundleOptimizer, CostMapExtractor,  # noqa: F401
                                     FeatureReferenceBundleOptimizer, ReferenceExtractor)


class BundleOptimizerOptions(dict):
    """Dict-constructible option struct of the reference; the optimizers take the dict itself."""


class ReferenceConfig(dict):
    pass


class CostMapConfig(dict):
    pass


def _outside(name, why):
    class _Outside:
        def __init__(self, *a, **k):
            raise NotImplementedError("%s is outside the accelerated path (%s)" % (name, why))
    _Outside.__name__ = name
    return _Outside


PatchWarpBundleOptimizer = _outside("PatchWarpBundleOptimizer", "N_NODES > 1 patch warping, DESIGN.md section 7")
GeometricBundleOptimizer = _outside("GeometricBundleOptimizer", "reprojection-error BA is COLMAP's, not featuremetric")

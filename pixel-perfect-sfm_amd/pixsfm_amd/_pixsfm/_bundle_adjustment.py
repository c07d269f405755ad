"""`_pixsfm._bundle_adjustment` (pixsfm/bundle_adjustment/bindings.cc:20-184)."""
from ..api.bundle_adjustment import (BundleAdjustmentSetup, CostMapBundleOptimizer, CostMapExtractor,  # noqa: F401
                                     FeatureReferenceBundleOptimizer, ReferenceExtractor)


from ._options import struct

BundleOptimizerOptions = struct("BundleOptimizerOptions", FeatureReferenceBundleOptimizer.option_defaults,
                                "BundleOptimizerOptions (bundle_adjustment_options.h:44-96; bindings.cc:113-135).")
ReferenceConfig = struct("ReferenceConfig", ReferenceExtractor.default_conf, "ReferenceConfig (reference_extractor.h:55-67; bindings.cc:69-79).")


def _effective_channels(self):
    """GetEffectiveChannels (costmap_extractor.h:52-61)."""
    return (4 if self["compute_cross_derivative"] else 3) if self["as_gradientfield"] else 1


CostMapConfig = struct("CostMapConfig", CostMapExtractor.default_conf, "CostMapConfig (costmap_extractor.h:38-62; bindings.cc:53-67).",
                       extra={"get_effective_channels": _effective_channels})
PatchWarpBundleOptimizerOptions = struct(
    "PatchWarpBundleOptimizerOptions", {**FeatureReferenceBundleOptimizer.option_defaults, "regularize_source": False},
    "Named by the bindings; its optimizer is outside the accelerated path.", base=BundleOptimizerOptions)


def _outside(name, why):
    class _Outside:
        def __init__(self, *a, **k):
            raise NotImplementedError("%s is outside the accelerated path (%s)" % (name, why))
    _Outside.__name__ = name
    return _Outside


PatchWarpBundleOptimizer = _outside("PatchWarpBundleOptimizer", "N_NODES > 1 patch warping, DESIGN.md section 7")
GeometricBundleOptimizer = _outside("GeometricBundleOptimizer", "reprojection-error BA is COLMAP's, not featuremetric")

"""`_pixsfm._keypoint_adjustment` (pixsfm/keypoint_adjustment/bindings.cc:17-81)."""
from ..api.keypoint_adjustment import (FeatureMetricKeypointOptimizer, KeypointAdjustmentSetup,  # noqa: F401
                                       TopologicalReferenceKeypointOptimizer)


class KeypointOptimizerOptions(dict):
    """Dict-constructible option struct of the reference (make_dataclass); the optimizers take the dict itself."""


FeatureMetricKeypointOptimizerOptions = KeypointOptimizerOptions

"""`_pixsfm._keypoint_adjustment` (pixsfm/keypoint_adjustment/bindings.cc:17-81)."""
from ..api.keypoint_adjustment import (FeatureMetricKeypointOptimizer, KeypointAdjustmentSetup,  # noqa: F401
                                       TopologicalReferenceKeypointOptimizer)


from ..api.keypoint_adjustment import FeatureMetricKeypointOptimizer as _FM
from ._options import struct

_FM_ONLY = ("num_threads", "root_regularize_weight", "weight_by_sim", "root_edges_only")
KeypointOptimizerOptions = struct(
    "KeypointOptimizerOptions", {k: v for k, v in _FM.option_defaults.items() if k not in _FM_ONLY},
    "KeypointOptimizerOptions (keypoint_adjustment_options.h:46-80; bindings.cc:49-58).")
FeatureMetricKeypointOptimizerOptions = struct(
    "FeatureMetricKeypointOptimizerOptions", _FM.option_defaults,
    "FeatureMetricKeypointOptimizer::Options (featuremetric_keypoint_optimizer.h:30-36, topological_keypoint_optimizer.h; bindings.cc:60-75).")
TopologicalReferenceKeypointOptimizerOptions = struct(
    "TopologicalReferenceKeypointOptimizerOptions",
    {**_FM.option_defaults, 'weight_by_sim': False, 'root_regularize_weight': 1.0, 'root_edges_only': True},
    "TopologicalReferenceKeypointOptimizer::Options (topological_reference_keypoint_optimizer.h:8-15; bindings.cc:85-92): the "
    "struct's constructor presets three values.", base=FeatureMetricKeypointOptimizerOptions)

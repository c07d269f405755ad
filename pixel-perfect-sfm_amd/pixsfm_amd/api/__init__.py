"""pixsfm-compatible Python surface of the accelerated KA/BA path (same class / method names as
pixsfm.keypoint_adjustment, pixsfm.bundle_adjustment, pixsfm._pixsfm._base/_features)."""
from . import base, features, localization, reconstruction  # noqa: F401
from .bundle_adjustment import (BundleAdjuster, BundleAdjustmentSetup, CostMapBundleAdjuster,  # noqa: F401
                                CostMapBundleOptimizer, CostMapExtractor, FeatureReferenceBundleAdjuster,
                                FeatureReferenceBundleOptimizer, FeatureView, ReferenceExtractor,
                                default_problem_setup)
from .keypoint_adjustment import (FeatureMetricKeypointAdjuster, FeatureMetricKeypointOptimizer,  # noqa: F401
                                  KeypointAdjuster, KeypointAdjustmentSetup,
                                  TopologicalReferenceKeypointAdjuster, TopologicalReferenceKeypointOptimizer,
                                  build_matching_graph, find_problem_labels)
from .localization import (QueryBundleAdjuster, QueryBundleOptimizer, QueryKeypointAdjuster,  # noqa: F401,E402
                           QueryKeypointOptimizer, find_feature_inliers, find_nearest_references)  # noqa: F401,E402

"""`_pixsfm._localization` (pixsfm/localization/bindings.cc)."""
from ..api.localization import QueryBundleOptimizer, QueryKeypointOptimizer, find_nearest_references  # noqa: F401


class QueryBundleOptimizerOptions(dict):
    pass

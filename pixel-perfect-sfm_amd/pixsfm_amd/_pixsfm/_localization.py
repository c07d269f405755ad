"""`_pixsfm._localization` (pixsfm/localization/bindings.cc)."""
from ..api.localization import QueryBundleOptimizer, QueryKeypointOptimizer, find_nearest_references  # noqa: F401


from ._options import struct

QueryBundleOptimizerOptions = struct("QueryBundleOptimizerOptions", QueryBundleOptimizer.option_defaults,
                                     "QueryBundleOptimizerOptions (query_refinement_options.h:8-57).")
QueryKeypointOptimizerOptions = struct("QueryKeypointOptimizerOptions", QueryKeypointOptimizer.option_defaults,
                                       "QueryKeypointOptimizerOptions (query_refinement_options.h:60-95).")

// Shadows features/src/featureset.h, featuremap.h, featureview.h (HDF5-backed containers): the names
// costmap_extractor.h's driver templates use, never instantiated by the shim.
#pragma once
#include <string>
#include <unordered_map>
#include "features/src/featurepatch.h"
namespace pixsfm {
template <typename dtype> class FeatureMap;
template <typename dtype> class FeatureSet;
template <typename dtype> class FeatureView;
}  // namespace pixsfm

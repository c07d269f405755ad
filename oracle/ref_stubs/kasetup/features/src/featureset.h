// Shadows features/src/featureset.h, featuremap.h, featureview.h (HDF5-backed containers) with in-memory ones that hold
// the patch METADATA the keypoint-adjustment set-up reads (corner, scale, shape): enough for the reference's
// ParameterizeKeypoints and for constructing (never evaluating) the cost functors.
#pragma once
#include <map>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>
#include "base/src/graph.h"
#include "features/src/featurepatch.h"
namespace pixsfm {
template <typename dtype>
class FeatureMap {
 public:
  bool IsSparse() const { return sparse; }
  int Channels() const { return channels; }
  int channels = 128;
  FeaturePatch<dtype>& GetFeaturePatch(colmap::point2D_t idx) { return *patches.at(idx); }
  std::unordered_map<colmap::point2D_t, FeaturePatch<dtype>*> patches;
  bool sparse = true;
};
template <typename dtype>
class FeatureSet {
 public:
  int Channels() const { return channels; }
  void FlushEveryN(int) {}
  void Flush() {}
  int channels = 128;
};
template <typename dtype>
class FeatureView {
 public:
  FeatureView() {}
  FeatureView(FeatureSet<dtype>*, const Graph*, const std::unordered_set<size_t>&) {}
  int Channels() const { return channels; }
  int channels = 128;
  FeatureMap<dtype>& GetFeatureMap(colmap::image_t image_id) { return maps.at(image_id); }
  // every residual the set-up code adds fetches its patch here first: the call log tells which observation a block is for
  FeaturePatch<dtype>& GetFeaturePatch(colmap::image_t image_id, colmap::point2D_t idx) {
    calls.emplace_back(image_id, idx);
    return maps.at(image_id).GetFeaturePatch(idx);
  }
  bool HasFeaturePatch(colmap::image_t image_id, colmap::point2D_t idx) const {
    auto it = maps.find(image_id);
    return it != maps.end() && it->second.patches.count(idx) > 0;
  }
  std::vector<std::pair<colmap::image_t, colmap::point2D_t>> calls;
  std::map<colmap::image_t, FeatureMap<dtype>> maps;
};
}  // namespace pixsfm

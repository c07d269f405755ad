// Shadows bundle_adjustment/src/reference_extractor.h: the two entry points costmap_extractor.h's drivers call.
#pragma once
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "features/src/references.h"
namespace pixsfm {
using Refs = std::unordered_map<colmap::point3D_t, Reference>;
class ReferenceExtractor {
 public:
  std::unordered_map<colmap::point3D_t, Reference> InitReferences(std::vector<int>&);
  template <int CHANNELS, int N_NODES, typename... A> double RunSubset(A&&...);
};
}  // namespace pixsfm

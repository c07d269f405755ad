// Shadows features/src/references.h (numpy / COLMAP track plumbing): the Reference FillPointCostmap reads its descriptor from.
#pragma once
#include <unordered_map>
#include "colmap/base/track.h"
#include "util/src/types.h"
namespace pixsfm {
struct Reference {
  DescriptorMatrixXd descriptor;
  double* DescriptorData() { return descriptor.data(); }
  const double* DescriptorData() const { return descriptor.data(); }
};
}  // namespace pixsfm

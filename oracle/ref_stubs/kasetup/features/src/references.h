#pragma once
#include <unordered_map>
#include "util/src/types.h"
namespace pixsfm {
struct Reference {
  DescriptorMatrixXd descriptor;
  const double* DescriptorData() const { return descriptor.data(); }
  double* DescriptorData() { return descriptor.data(); }
  const double* NodeOffsets3DData() const { return nullptr; }
};
}  // namespace pixsfm

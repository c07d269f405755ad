// Stub shadowing the reference's base/src/interpolation.h (which needs the real Eigen / Ceres / pybind11) for the
// in-place build of base/src/irls_optim.h: only the two InterpolationConfig fields and the NCCNormalize symbol that
// RobustMeanIRLS touches (interpolation.h:39-51).  ncc_normalize is outside the hot path (N_NODES = 1) and stays false.
#pragma once
#include <array>
#include <cstdlib>
#include <vector>
namespace pixsfm {
struct InterpolationConfig {
  bool l2_normalize = true;
  bool ncc_normalize = false;
  std::vector<std::array<double, 2>> nodes = {{0.0, 0.0}};
};
template <typename T>
inline void NCCNormalize(T*, int, int) { std::abort(); }
}  // namespace pixsfm

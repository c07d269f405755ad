// Shadows util/src/statistics.h (summary merging / printing): declarations for the optimizer's reporting code.
#pragma once
#include <vector>
#include <ceres/ceres.h>
namespace pixsfm {
template <typename T> ceres::Solver::Summary AccumulateSummaries(const T&) { return ceres::Solver::Summary(); }
inline void PrintSolverSummary(const ceres::Solver::Summary&) {}
}  // namespace pixsfm

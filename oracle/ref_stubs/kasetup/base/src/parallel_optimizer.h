// Shadows base/src/parallel_optimizer.h (thread pool over sub-problems): the base class the optimizers name.
#pragma once
#include <vector>
#include <ceres/ceres.h>
namespace pixsfm {
template <typename Derived, typename idx_t>
class ParallelOptimizer {
 public:
  explicit ParallelOptimizer(int n_threads) : n_threads_(n_threads) {}
  template <typename... A> std::vector<ceres::Solver::Summary> RunParallel(A&&...) { return std::vector<ceres::Solver::Summary>(); }
 protected:
  int n_threads_; double parallel_solver_time_ = 0.0;
};
}  // namespace pixsfm

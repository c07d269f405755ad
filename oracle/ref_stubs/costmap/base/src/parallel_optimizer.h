// Shadows base/src/parallel_optimizer.h (thread pool over sub-problems): the base class CostMapExtractor names.
#pragma once
#include <vector>
namespace pixsfm {
template <typename Derived, typename idx_t>
class ParallelOptimizer {
 public:
  explicit ParallelOptimizer(int n_threads) : n_threads_(n_threads) {}
  template <typename... A> std::vector<double> RunParallel(A&&...) { return std::vector<double>(); }
 protected:
  int n_threads_; double parallel_solver_time_ = 0.0;
};
template <typename T> double AccumulateValues(const T&) { return 0.0; }
}  // namespace pixsfm

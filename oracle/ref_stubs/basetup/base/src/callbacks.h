// Shadows base/src/callbacks.h (Python interrupt + progress bar plumbing).
#pragma once
#include <cstddef>
#include <ceres/ceres.h>
namespace pixsfm {
struct PyInterruptCallback : public ceres::IterationCallback {};
struct StubBar { void update(int = 0) {} void finish() {} };
struct ProgressBarIterationCallback : public ceres::IterationCallback {
  explicit ProgressBarIterationCallback(size_t) {}
  StubBar& ProgressBar() { return bar_; }
  StubBar bar_;
};
}  // namespace pixsfm

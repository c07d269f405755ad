// Shadows base/src/callbacks.h (Python keyboard-interrupt plumbing): the callback type the optimizer registers.
#pragma once
#include <ceres/ceres.h>
namespace pixsfm {
struct PyInterruptCallback : public ceres::IterationCallback {};
}  // namespace pixsfm

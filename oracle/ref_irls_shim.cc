// ref_irls_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN robust-mean loop where it lies under /root/reference:
//   pixsfm/base/src/irls_optim.h   (RobustMeanIRLS, what ReferenceExtractor::ComputeReference runs per 3D point)
// against the stub headers in oracle/ref_stubs/irls/ (a minimal matrix class instead of Eigen, the
// ceres::LossFunction interface, empty pybind11 / HighFive / COLMAP headers).  Output:
// oracle/_ref/libpxo_ref_irls.so.  Nothing of the reference is copied into this repository.
#include <cstdint>
#include <memory>
#include <vector>

#include "base/src/irls_optim.h"

extern "C" {

// descs: n x C row-major; loss_type 0 trivial, 1 cauchy, 2 huber (scale a).  out_mean: C doubles.
// Returns 0, or 1 when the reference returned early with one of the observations (a loss value <= 0).
int pxo_ref_robust_mean_irls(const double* descs, int n, int C, int loss_type, double a, int iters, int l2_normalize,
                             double* out_mean) {
  std::vector<pixsfm::DescriptorMatrixd<Eigen::Dynamic, Eigen::Dynamic>> track;
  for (int i = 0; i < n; ++i) {
    Eigen::MiniMat d(1, C);
    for (int c = 0; c < C; ++c) d.data()[c] = descs[(size_t)i * C + c];
    track.push_back(d);
  }
  std::unique_ptr<ceres::LossFunction> loss;
  if (loss_type == 0) loss.reset(new ceres::TrivialLoss());
  else if (loss_type == 1) loss.reset(new ceres::CauchyLoss(a));
  else loss.reset(new ceres::HuberLoss(a));
  pixsfm::InterpolationConfig cfg;
  cfg.l2_normalize = l2_normalize != 0;
  Eigen::MiniMat mean = pixsfm::RobustMeanIRLS<Eigen::Dynamic, Eigen::Dynamic>(track, loss.get(), iters, cfg);
  int early = 0;
  for (int i = 0; i < n && !early; ++i) {
    bool same = true;
    for (int c = 0; c < C; ++c) same = same && mean.data()[c] == descs[(size_t)i * C + c];
    if (same) early = 1;
  }
  for (int c = 0; c < C; ++c) out_mean[c] = mean.data()[c];
  return early;
}

}  // extern "C"

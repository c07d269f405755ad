// ref_costmap_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN cost-map kernel where it lies under /root/reference:
//   pixsfm/bundle_adjustment/src/costmap_extractor.h   CostMapExtractor::FillPointCostmap (:230-358): both branches (raw texels
//       with storage-type central differences; PatchInterpolator::EvaluateLocal for upsampled maps / the cross derivative),
//       the loss, the `cost > 1e-8` gate, the sqrt variants, FeaturePatch::SetEntry's cast to the storage type
// on top of features/src/featurepatch.h, patch_interpolator.h, base/src/interpolation.h and third-party/half.hpp (all real),
// against oracle/ref_stubs/costmap/ (shadows of the HDF5 / COLMAP containers the extractor's DRIVER templates name; they are
// never instantiated) and oracle/ref_stubs/interp/ (matrix class, loss functions restated from the published Ceres formulas).
// Output: oracle/_ref/libpxo_ref_costmap.so.  Nothing of the reference is copied into this repository.
#include <array>
#include <cstdint>
#include <memory>

#include "bundle_adjustment/src/costmap_extractor.h"

namespace pixsfm {
template <typename dtype>
FeaturePatch<dtype>::FeaturePatch() : data_ptr_(nullptr) {}

template <typename dtype>
struct ViewPatch : public FeaturePatch<dtype> {
  ViewPatch(void* data, int H, int W, int C) {
    this->data_ptr_ = static_cast<dtype*>(data);
    this->shape_ = {H, W, C};
    this->corner_[0] = 0; this->corner_[1] = 0;
    this->scale_[0] = 1.0; this->scale_[1] = 1.0;
  }
};

struct Filler : public CostMapExtractor {
  using CostMapExtractor::CostMapExtractor;
  template <int CHANNELS, typename dtype_o, typename dtype>
  void Fill(void* feat, int H, int W, const double* ref, void* out, int Ho, int Wo, int CO) {
    ViewPatch<dtype> fpatch(feat, H, W, CHANNELS);
    ViewPatch<dtype_o> cost(out, Ho, Wo, CO);
    Reference reference;
    reference.descriptor = DescriptorMatrixXd(1, CHANNELS);
    for (int i = 0; i < CHANNELS; ++i) reference.descriptor.data()[i] = ref[i];
    this->template FillPointCostmap<CHANNELS, dtype_o, dtype>(fpatch, reference, cost);
  }
};
}  // namespace pixsfm

extern "C" {
// feat: H x W x 128 patch of dtype (0 half, 1 float, 2 double); out: Ho x Wo x CO cost map of out_dtype, CO = 1 (plain), 3
// (gradient field) or 4 (+ cross derivative).  loss_type 0 trivial, 1 cauchy(a), 2 huber(a).  Returns 0, -1 when the dtype
// pair is not instantiated.
int pxo_ref_fill_point_costmap(void* feat, int dtype, int H, int W, const double* ref, void* out, int out_dtype, int Ho, int Wo,
                               double upsampling_factor, int as_gradientfield, int compute_cross_derivative, int apply_sqrt,
                               int loss_type, double a, int l2_normalize) {
  pixsfm::CostMapConfig cfg;
  cfg.upsampling_factor = upsampling_factor;
  cfg.as_gradientfield = as_gradientfield != 0;
  cfg.compute_cross_derivative = compute_cross_derivative != 0;
  cfg.apply_sqrt = apply_sqrt != 0;
  if (loss_type == 1) cfg.loss.reset(new ceres::CauchyLoss(a));
  else if (loss_type == 2) cfg.loss.reset(new ceres::HuberLoss(a));
  pixsfm::InterpolationConfig icfg;
  icfg.l2_normalize = l2_normalize != 0;
  pixsfm::Filler filler(cfg, icfg);
  const int CO = cfg.GetEffectiveChannels();
#define FILL(CH, DO, DI) { filler.Fill<CH, DO, DI>(feat, H, W, ref, out, Ho, Wo, CO); return 0; }
  if (dtype == 0 && out_dtype == 0) FILL(128, half, half)
  if (dtype == 0 && out_dtype == 1) FILL(128, float, half)
  if (dtype == 0 && out_dtype == 2) FILL(128, double, half)
  if (dtype == 1 && out_dtype == 1) FILL(128, float, float)
  if (dtype == 2 && out_dtype == 2) FILL(128, double, double)
  return -1;
}

// The extractor's other registered case, CHANNELS = 3 (costmap_extractor.h:35-37: image intensities): feat is H x W x 3.  Only
// the branch WITHOUT interpolation (cost map of the patch's size, no cross derivative) -- below 8 channels the interpolating
// branch runs on ceres::BiCubicInterpolator, which is not available here.
int pxo_ref_fill_point_costmap3(void* feat, int dtype, int H, int W, const double* ref, void* out, int out_dtype, int as_gradientfield,
                                int apply_sqrt, int loss_type, double a) {
  pixsfm::CostMapConfig cfg;
  cfg.as_gradientfield = as_gradientfield != 0;
  cfg.apply_sqrt = apply_sqrt != 0;
  if (loss_type == 1) cfg.loss.reset(new ceres::CauchyLoss(a));
  else if (loss_type == 2) cfg.loss.reset(new ceres::HuberLoss(a));
  pixsfm::InterpolationConfig icfg;
  icfg.l2_normalize = false;
  pixsfm::Filler filler(cfg, icfg);
  const int CO = cfg.GetEffectiveChannels();
  const int Ho = H, Wo = W;
  if (dtype == 0 && out_dtype == 0) FILL(3, half, half)
  if (dtype == 0 && out_dtype == 1) FILL(3, float, half)
  if (dtype == 0 && out_dtype == 2) FILL(3, double, half)
  if (dtype == 1 && out_dtype == 1) FILL(3, float, float)
  if (dtype == 2 && out_dtype == 2) FILL(3, double, double)
#undef FILL
  return -1;
}
}  // extern "C"

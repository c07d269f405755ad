// ref_interp_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN interpolation stack where it lies under /root/reference:
//   pixsfm/base/src/interpolation.h            (BiCubicInterpolator::EvaluateSIMD, PixelInterpolator: L2 normalisation and
//                                               its chain rule, the Jet bridge  f.v = dfdr r.v + dfdc c.v)
//   pixsfm/base/src/cubic_hermite_spline_simd.h, base/src/grid2d.h, third-party/half.hpp
//   pixsfm/features/src/featurepatch.h         (ToPixelCoordinates: scale, the half-pixel shift, corner, upsampling factor)
//   pixsfm/features/src/patch_interpolator.h   (Evaluate / EvaluateLocal / CheckBounds) and util/src/math.h (IsInsideZeroL)
// against the stub headers in oracle/ref_stubs/interp/ (a small dense matrix instead of Eigen, a dual number instead of
// ceres::Jet, empty pybind11 / HighFive / Boost / COLMAP headers).  Output: oracle/_ref/libpxo_ref_interp.so.
// Nothing of the reference is copied into this repository.
//
// What the shim adds: a patch that points at caller memory (FeaturePatch's constructors live in featurepatch.cc together
// with the HDF5 / numpy loaders, which are not built) and the seeding of the two dual numbers exactly as
// ceres::AutoDiffCostFunction<..., 2> seeds a 2-parameter block.
#include <array>
#include <cstdint>

#include "features/src/featurepatch.h"
#include "features/src/patch_interpolator.h"

namespace pixsfm {
template <typename dtype>
FeaturePatch<dtype>::FeaturePatch() : data_ptr_(nullptr) {}

template <typename dtype>
struct ViewPatch : public FeaturePatch<dtype> {
  ViewPatch(const void* data, int H, int W, int C, int cx, int cy, double sx, double sy, double up) {
    this->data_ptr_ = const_cast<dtype*>(static_cast<const dtype*>(data));
    this->shape_ = {H, W, C};
    this->corner_[0] = cx; this->corner_[1] = cy;
    this->scale_[0] = sx; this->scale_[1] = sy;
    this->upsampling_factor_ = up;
  }
};
}  // namespace pixsfm

namespace {
using pixsfm::InterpolationConfig;

template <typename dtype, int C>
int PatchEval(const void* data, int H, int W, int cx, int cy, double sx, double sy, double up, const InterpolationConfig& cfg,
              const double* xy, double* f, double* gx, double* gy) {
  pixsfm::ViewPatch<dtype> patch(data, H, W, C, cx, cy, sx, sy, up);
  pixsfm::PatchInterpolator<dtype, C> interp(cfg, patch);
  if (!gx) {
    return interp.template Evaluate<double>(xy, f) ? 1 : 0;
  }
  typedef ceres::Jet<double, 2> J;
  J in[2] = {J(xy[0], 0), J(xy[1], 1)};
  std::vector<J> out(C);
  const bool inside = interp.template Evaluate<J>(in, out.data());
  for (int i = 0; i < C; ++i) { f[i] = out[i].a; gx[i] = out[i].v[0]; gy[i] = out[i].v[1]; }
  return inside ? 1 : 0;
}

template <typename dtype, int C>
int LocalEval(const void* data, int H, int W, const InterpolationConfig& cfg, double* xy, double* f, double* dfdr, double* dfdc,
              double* dfdrc) {
  pixsfm::ViewPatch<dtype> patch(data, H, W, C, 0, 0, 1.0, 1.0, 1.0);
  pixsfm::PatchInterpolator<dtype, C> interp(cfg, patch);
  return interp.EvaluateLocal(xy, f, dfdr, dfdc, dfdrc) ? 1 : 0;
}

InterpolationConfig MakeCfg(int l2_normalize, int use_float_simd, int check_bounds) {
  InterpolationConfig cfg;
  cfg.l2_normalize = l2_normalize != 0;
  cfg.use_float_simd = use_float_simd != 0;
  cfg.check_bounds = check_bounds != 0;
  return cfg;
}
}  // namespace

extern "C" {

// dtype: 0 = half, 1 = float, 2 = double; C in {128, 64}.  xy: keypoint in image coordinates (COLMAP convention).
// f / gx / gy: descriptor and its derivatives with respect to x and y (gx == NULL: value only).  Returns is_inside (0 / 1),
// -1 for a combination that is not instantiated.
int pxo_ref_patch_eval(const void* data, int dtype, int H, int W, int C, int cx, int cy, double sx, double sy, double up,
                       int l2_normalize, int use_float_simd, int check_bounds, const double* xy, double* f, double* gx,
                       double* gy) {
  const InterpolationConfig cfg = MakeCfg(l2_normalize, use_float_simd, check_bounds);
#define PE(DT, CC) return PatchEval<DT, CC>(data, H, W, cx, cy, sx, sy, up, cfg, xy, f, gx, gy)
  if (C == 128) { if (dtype == 0) PE(half, 128); if (dtype == 1) PE(float, 128); if (dtype == 2) PE(double, 128); }
  if (C == 64) { if (dtype == 0) PE(half, 64); if (dtype == 1) PE(float, 64); if (dtype == 2) PE(double, 64); }
#undef PE
  return -1;
}

// PatchInterpolator::EvaluateLocal in LOCAL patch coordinates xy = (column, row): f, df/dr, df/dc and (dfdrc != NULL) the
// cross derivative -- the path CostMapExtractor's interpolating branch takes.
int pxo_ref_patch_eval_local(const void* data, int dtype, int H, int W, int C, int l2_normalize, int use_float_simd,
                             int check_bounds, const double* xy, double* f, double* dfdr, double* dfdc, double* dfdrc) {
  const InterpolationConfig cfg = MakeCfg(l2_normalize, use_float_simd, check_bounds);
  double q[2] = {xy[0], xy[1]};
#define LE(DT, CC) return LocalEval<DT, CC>(data, H, W, cfg, q, f, dfdr, dfdc, dfdrc)
  if (C == 128) { if (dtype == 0) LE(half, 128); if (dtype == 1) LE(float, 128); if (dtype == 2) LE(double, 128); }
  if (C == 64) { if (dtype == 0) LE(half, 64); if (dtype == 1) LE(float, 64); if (dtype == 2) LE(double, 64); }
#undef LE
  return -1;
}

}  // extern "C"

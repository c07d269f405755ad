// ref_residual_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN residual functors where they lie under /root/reference and differentiates them the way
// ceres::AutoDiffCostFunction does (one dual number per parameter, blocks in declaration order):
//   pixsfm/residuals/src/featuremetric.h      FeatureMetric2DCostFunctor         (KA edge residual)
//   pixsfm/residuals/src/feature_reference.h  FeatureReference2DCostFunctor      (unary reference residual, A8)
//                                             FeatureReferenceCostFunctor        (BA residual, A9)
//                                             FeatureReferenceConstantPoseCostFunctor (A10)
//   pixsfm/base/src/projection.h              WorldToPixel  (rotate, translate, divide, CameraModel::WorldToImage)
// on top of the interpolation stack of ref_interp_shim.cc, against the stub headers in oracle/ref_stubs/interp/.  What is
// NOT reference code underneath: ceres::QuaternionRotatePoint, the COLMAP camera models and the dual number itself
// (ref_stubs/interp/ceres/rotation.h, colmap/base/camera_models.h, ceres/ceres.h: restated from the published upstream
// definitions).  Output: oracle/_ref/libpxo_ref_residual.so.  Nothing of the reference is copied into this repository.
#include <array>
#include <cstdint>
#include <vector>

#include "residuals/src/feature_reference.h"
#include "residuals/src/featuremetric.h"

namespace pixsfm {
template <typename dtype>
FeaturePatch<dtype>::FeaturePatch() : data_ptr_(nullptr) {}

template <typename dtype>
struct ViewPatch : public FeaturePatch<dtype> {
  ViewPatch(const void* data, int H, int W, int C, const int* corner, const double* scale) {
    this->data_ptr_ = const_cast<dtype*>(static_cast<const dtype*>(data));
    this->shape_ = {H, W, C};
    this->corner_[0] = corner[0]; this->corner_[1] = corner[1];
    this->scale_[0] = scale[0]; this->scale_[1] = scale[1];
  }
};
}  // namespace pixsfm

namespace {
using pixsfm::InterpolationConfig;

InterpolationConfig MakeCfg(int l2_normalize, int use_float_simd, int check_bounds) {
  InterpolationConfig cfg;
  cfg.l2_normalize = l2_normalize != 0;
  cfg.use_float_simd = use_float_simd != 0;
  cfg.check_bounds = check_bounds != 0;
  return cfg;
}

template <int N>
void Seed(const double* x, int n, int first, ceres::Jet<double, N>* out) {
  for (int i = 0; i < n; ++i) out[i] = ceres::Jet<double, N>(x[i], first + i);
}

template <typename dtype, int C>
int KaEdge(const void* d1, const void* d2, int H, int W, const int* c1, const double* s1, const int* c2, const double* s2,
           InterpolationConfig& cfg, const double* kp1, const double* kp2, double* r, double* J1, double* J2) {
  pixsfm::ViewPatch<dtype> p1(d1, H, W, C, c1, s1), p2(d2, H, W, C, c2, s2);
  pixsfm::FeatureMetric2DCostFunctor<dtype, C, 1> functor(p1, p2, cfg);
  typedef ceres::Jet<double, 4> J;
  J a[2], b[2];
  Seed<4>(kp1, 2, 0, a); Seed<4>(kp2, 2, 2, b);
  std::vector<J> out(C);
  const bool ok = functor(a, b, out.data());
  for (int i = 0; i < C; ++i) {
    r[i] = out[i].a;
    J1[2 * i] = out[i].v[0]; J1[2 * i + 1] = out[i].v[1];
    J2[2 * i] = out[i].v[2]; J2[2 * i + 1] = out[i].v[3];
  }
  return ok ? 1 : 0;
}

template <typename dtype, int C>
int Ref2D(const void* d, int H, int W, const int* c, const double* s, InterpolationConfig& cfg, const double* kp,
          const double* ref, double* r, double* Jk) {
  pixsfm::ViewPatch<dtype> p(d, H, W, C, c, s);
  pixsfm::FeatureReference2DCostFunctor<dtype, C, 1> functor(p, cfg, ref);
  typedef ceres::Jet<double, 2> J;
  J a[2];
  Seed<2>(kp, 2, 0, a);
  std::vector<J> out(C);
  const bool ok = functor(a, out.data());
  for (int i = 0; i < C; ++i) { r[i] = out[i].a; Jk[2 * i] = out[i].v[0]; Jk[2 * i + 1] = out[i].v[1]; }
  return ok ? 1 : 0;
}

// J: C x (10 + K) row-major, columns qvec (4) | tvec (3) | point (3) | camera parameters (K)
template <typename CameraModel, typename dtype, int C>
int BaResidual(const void* d, int H, int W, const int* c, const double* s, InterpolationConfig& cfg, const double* q,
               const double* t, const double* X, const double* params, const double* ref, double* r, double* Jout) {
  constexpr int K = (int)CameraModel::kNumParams, N = 10 + K;
  pixsfm::ViewPatch<dtype> p(d, H, W, C, c, s);
  pixsfm::FeatureReferenceCostFunctor<CameraModel, dtype, C, 1> functor(p, cfg, ref);
  typedef ceres::Jet<double, N> J;
  J jq[4], jt[3], jX[3], jk[K];
  Seed<N>(q, 4, 0, jq); Seed<N>(t, 3, 4, jt); Seed<N>(X, 3, 7, jX); Seed<N>(params, K, 10, jk);
  std::vector<J> out(C);
  const bool ok = functor(jq, jt, jX, jk, out.data());
  for (int i = 0; i < C; ++i) {
    r[i] = out[i].a;
    for (int k = 0; k < N; ++k) Jout[(size_t)i * N + k] = out[i].v[k];
  }
  return ok ? 1 : 0;
}

// constant pose: J is C x (3 + K), columns point (3) | camera parameters (K)
template <typename CameraModel, typename dtype, int C>
int BaResidualConstPose(const void* d, int H, int W, const int* c, const double* s, InterpolationConfig& cfg, const double* q,
                        const double* t, const double* X, const double* params, const double* ref, double* r, double* Jout) {
  constexpr int K = (int)CameraModel::kNumParams, N = 3 + K;
  pixsfm::ViewPatch<dtype> p(d, H, W, C, c, s);
  pixsfm::FeatureReferenceConstantPoseCostFunctor<CameraModel, dtype, C, 1> functor(p, cfg, q, t, ref);
  typedef ceres::Jet<double, N> J;
  J jX[3], jk[K];
  Seed<N>(X, 3, 0, jX); Seed<N>(params, K, 3, jk);
  std::vector<J> out(C);
  const bool ok = functor(jX, jk, out.data());
  for (int i = 0; i < C; ++i) {
    r[i] = out[i].a;
    for (int k = 0; k < N; ++k) Jout[(size_t)i * N + k] = out[i].v[k];
  }
  return ok ? 1 : 0;
}
}  // namespace

extern "C" {
// All patches: C = 128 channels; dtype 0 = half, 1 = float, 2 = double.  Return: the functor's bool (1 / 0), -1 when the
// combination is not instantiated.

int pxo_ref_ka_edge(const void* d1, const void* d2, int dtype, int H, int W, const int* c1, const double* s1, const int* c2,
                    const double* s2, int l2, int fs, int cb, const double* kp1, const double* kp2, double* r, double* J1,
                    double* J2) {
  InterpolationConfig cfg = MakeCfg(l2, fs, cb);
  if (dtype == 0) return KaEdge<half, 128>(d1, d2, H, W, c1, s1, c2, s2, cfg, kp1, kp2, r, J1, J2);
  if (dtype == 1) return KaEdge<float, 128>(d1, d2, H, W, c1, s1, c2, s2, cfg, kp1, kp2, r, J1, J2);
  if (dtype == 2) return KaEdge<double, 128>(d1, d2, H, W, c1, s1, c2, s2, cfg, kp1, kp2, r, J1, J2);
  return -1;
}

int pxo_ref_ref2d(const void* d, int dtype, int H, int W, const int* c, const double* s, int l2, int fs, int cb,
                  const double* kp, const double* ref, double* r, double* J) {
  InterpolationConfig cfg = MakeCfg(l2, fs, cb);
  if (dtype == 0) return Ref2D<half, 128>(d, H, W, c, s, cfg, kp, ref, r, J);
  if (dtype == 1) return Ref2D<float, 128>(d, H, W, c, s, cfg, kp, ref, r, J);
  if (dtype == 2) return Ref2D<double, 128>(d, H, W, c, s, cfg, kp, ref, r, J);
  return -1;
}

// model: COLMAP model id 0 .. 4 (SIMPLE_PINHOLE, PINHOLE, SIMPLE_RADIAL, RADIAL, OPENCV); half patches.  const_pose != 0:
// FeatureReferenceConstantPoseCostFunctor.
int pxo_ref_ba_residual(int model, int const_pose, const void* d, int H, int W, const int* c, const double* s, int l2, int fs,
                        int cb, const double* q, const double* t, const double* X, const double* params, const double* ref,
                        double* r, double* J) {
  InterpolationConfig cfg = MakeCfg(l2, fs, cb);
#define BA(M) return const_pose ? BaResidualConstPose<colmap::M, half, 128>(d, H, W, c, s, cfg, q, t, X, params, ref, r, J) \
                                : BaResidual<colmap::M, half, 128>(d, H, W, c, s, cfg, q, t, X, params, ref, r, J)
  switch (model) {
    case 0: BA(SimplePinholeCameraModel);
    case 1: BA(PinholeCameraModel);
    case 2: BA(SimpleRadialCameraModel);
    case 3: BA(RadialCameraModel);
    case 4: BA(OpenCVCameraModel);
    default: return -1;
  }
#undef BA
}
}  // extern "C"

// Stub for <colmap/base/camera_models.h>: the camera models the residual headers name, with WorldToImage / Distortion
// restated from their published definitions ([upstream COLMAP 3.8] src/base/camera_models.h).  NOT reference code: parity
// of a projection rests on this stub as far as the camera model itself goes (SURVEY 8a row A6 stays unpinned).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <vector>
namespace colmap {
struct SimplePinholeCameraModel {
  static const int kModelId = 0; static const size_t kNumParams = 3;
  template <typename T> static void Distortion(const T*, const T, const T, T* du, T* dv) { *du = T(0); *dv = T(0); }
  template <typename T> static void WorldToImage(const T* p, const T u, const T v, T* x, T* y) { *x = p[0] * u + p[1]; *y = p[0] * v + p[2]; }
};
struct PinholeCameraModel {
  static const int kModelId = 1; static const size_t kNumParams = 4;
  template <typename T> static void Distortion(const T*, const T, const T, T* du, T* dv) { *du = T(0); *dv = T(0); }
  template <typename T> static void WorldToImage(const T* p, const T u, const T v, T* x, T* y) { *x = p[0] * u + p[2]; *y = p[1] * v + p[3]; }
};
struct SimpleRadialCameraModel {
  static const int kModelId = 2; static const size_t kNumParams = 4;
  template <typename T> static void Distortion(const T* e, const T u, const T v, T* du, T* dv) {
    const T k = e[0]; const T u2 = u * u; const T v2 = v * v; const T r2 = u2 + v2; const T radial = k * r2; *du = u * radial; *dv = v * radial;
  }
  template <typename T> static void WorldToImage(const T* p, const T u, const T v, T* x, T* y) {
    const T f = p[0]; const T c1 = p[1]; const T c2 = p[2];
    T du, dv; Distortion(&p[3], u, v, &du, &dv);
    *x = u + du; *y = v + dv;
    *x = f * *x + c1; *y = f * *y + c2;
  }
};
struct RadialCameraModel {
  static const int kModelId = 3; static const size_t kNumParams = 5;
  template <typename T> static void Distortion(const T* e, const T u, const T v, T* du, T* dv) {
    const T k1 = e[0]; const T k2 = e[1]; const T u2 = u * u; const T v2 = v * v; const T r2 = u2 + v2; const T radial = k1 * r2 + k2 * r2 * r2;
    *du = u * radial; *dv = v * radial;
  }
  template <typename T> static void WorldToImage(const T* p, const T u, const T v, T* x, T* y) {
    const T f = p[0]; const T c1 = p[1]; const T c2 = p[2];
    T du, dv; Distortion(&p[3], u, v, &du, &dv);
    *x = u + du; *y = v + dv;
    *x = f * *x + c1; *y = f * *y + c2;
  }
};
struct OpenCVCameraModel {
  static const int kModelId = 4; static const size_t kNumParams = 8;
  template <typename T> static void Distortion(const T* e, const T u, const T v, T* du, T* dv) {
    const T k1 = e[0]; const T k2 = e[1]; const T p1 = e[2]; const T p2 = e[3];
    const T u2 = u * u; const T uv = u * v; const T v2 = v * v; const T r2 = u2 + v2; const T radial = k1 * r2 + k2 * r2 * r2;
    *du = u * radial + T(2) * p1 * uv + p2 * (r2 + T(2) * u2);
    *dv = v * radial + T(2) * p2 * uv + p1 * (r2 + T(2) * v2);
  }
  template <typename T> static void WorldToImage(const T* p, const T u, const T v, T* x, T* y) {
    const T f1 = p[0]; const T f2 = p[1]; const T c1 = p[2]; const T c2 = p[3];
    T du, dv; Distortion(&p[4], u, v, &du, &dv);
    *x = u + du; *y = v + dv;
    *x = f1 * *x + c1; *y = f2 * *y + c2;
  }
};
class Camera {
 public:
  int ModelId() const { return model_; }
  const double* ParamsData() const { return params_.data(); }
  double* ParamsData() { return params_.data(); }
  const std::vector<double>& Params() const { return params_; }
  size_t NumParams() const { return params_.size(); }
  void SetModelId(int m) { model_ = m; }
  void SetParams(const std::vector<double>& p) { params_ = p; }
  template <typename V> V ImageToWorld(const V&) const { std::fprintf(stderr, "colmap stub: Camera::ImageToWorld is not available\n"); std::abort(); }
  // [upstream COLMAP 3.8 camera_models.h] parameter groups: focal length(s), principal point, extra (distortion) parameters
  std::vector<size_t> FocalLengthIdxs() const { return (model_ == 1 || model_ == 4) ? std::vector<size_t>{0, 1} : std::vector<size_t>{0}; }
  std::vector<size_t> PrincipalPointIdxs() const { return (model_ == 1 || model_ == 4) ? std::vector<size_t>{2, 3} : std::vector<size_t>{1, 2}; }
  std::vector<size_t> ExtraParamsIdxs() const {
    switch (model_) { case 2: return {3}; case 3: return {3, 4}; case 4: return {4, 5, 6, 7}; default: return {}; }
  }
 private:
  int model_ = 0; std::vector<double> params_;
};
}  // namespace colmap
#define CAMERA_MODEL_CASES CAMERA_MODEL_CASE(SimplePinholeCameraModel) CAMERA_MODEL_CASE(PinholeCameraModel) \
  CAMERA_MODEL_CASE(SimpleRadialCameraModel) CAMERA_MODEL_CASE(RadialCameraModel) CAMERA_MODEL_CASE(OpenCVCameraModel)
#define CAMERA_MODEL_DOES_NOT_EXIST_EXCEPTION default: throw std::domain_error("Camera model does not exist"); break;
#define CAMERA_MODEL_SWITCH_CASES CAMERA_MODEL_CASES CAMERA_MODEL_DOES_NOT_EXIST_EXCEPTION

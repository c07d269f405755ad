// pxr_camera_ext.h -- the less common COLMAP 3.8 camera models of CAMERA_MODEL_SWITCH_CASES
// (residuals/src/feature_reference.h:232 -> [upstream colmap/base/camera_models.h]):
// OPENCV_FISHEYE (5), FULL_OPENCV (6), FOV (7), SIMPLE_RADIAL_FISHEYE (8), RADIAL_FISHEYE (9),
// THIN_PRISM_FISHEYE (10).  One templated WorldToImage serves both the value path (T = double,
// residual kernel prologue) and the Jacobian path (T = forward-mode dual number with 2 + K
// partials, per-observation linearisation kernels -- the reference differentiates the same
// formulas with ceres::Jet).  The five common models keep their hand-derived Jacobians in
// pxr_device.h.
#pragma once
#include <hip/hip_runtime.h>

#include "pixsfm_hip.h"

namespace pxr {

template <int N>
struct Dual {
  double a;
  double v[N];
  __device__ __forceinline__ Dual() {}
  __device__ __forceinline__ Dual(double x) : a(x) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 0.0;
  }
};
#define PXR_DUAL_BIN(op, expr_a, expr_v)                                                      \
  template <int N> __device__ __forceinline__ Dual<N> operator op(const Dual<N>& x, const Dual<N>& y) { \
    Dual<N> r; r.a = expr_a;                                                                   \
    _Pragma("unroll") for (int i = 0; i < N; ++i) r.v[i] = expr_v;                             \
    return r; }
PXR_DUAL_BIN(+, x.a + y.a, x.v[i] + y.v[i])
PXR_DUAL_BIN(-, x.a - y.a, x.v[i] - y.v[i])
PXR_DUAL_BIN(*, x.a * y.a, x.a * y.v[i] + x.v[i] * y.a)
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& x, const Dual<N>& y) {
  Dual<N> r; const double iy = 1.0 / y.a; r.a = x.a * iy;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * iy;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& x, double c) { Dual<N> r = x; r.a += c; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(double c, const Dual<N>& x) { return x + c; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& x, double c) { Dual<N> r = x; r.a -= c; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& x, double c) {
  Dual<N> r; r.a = x.a * c;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * c;
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator*(double c, const Dual<N>& x) { return x * c; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& x, double c) { return x * (1.0 / c); }
template <int N> __device__ __forceinline__ Dual<N> chain(const Dual<N>& x, double f, double df) {
  Dual<N> r; r.a = f;
#pragma unroll
  for (int i = 0; i < N; ++i) r.v[i] = df * x.v[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> dsqrt(const Dual<N>& x) { const double s = sqrt(x.a); return chain(x, s, 0.5 / s); }
template <int N> __device__ __forceinline__ Dual<N> datan(const Dual<N>& x) { return chain(x, atan(x.a), 1.0 / (1.0 + x.a * x.a)); }
template <int N> __device__ __forceinline__ Dual<N> dtan(const Dual<N>& x) { const double t = tan(x.a); return chain(x, t, 1.0 + t * t); }
__device__ __forceinline__ double dsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double datan(double x) { return atan(x); }
__device__ __forceinline__ double dtan(double x) { return tan(x); }
__device__ __forceinline__ double real_part(double x) { return x; }
template <int N> __device__ __forceinline__ double real_part(const Dual<N>& x) { return x.a; }

__device__ __forceinline__ int camera_num_params_ext(int model) {
  switch (model) {
    case PXR_OPENCV_FISHEYE: return 8;
    case PXR_FULL_OPENCV: return 12;
    case PXR_FOV: return 5;
    case PXR_SIMPLE_RADIAL_FISHEYE: return 4;
    case PXR_RADIAL_FISHEYE: return 5;
    case PXR_THIN_PRISM_FISHEYE: return 12;
    default: return 0;
  }
}

// [upstream COLMAP 3.8] <Model>::WorldToImage for the six models above.  k: PXR_KPAD parameters.
template <typename T>
__device__ inline bool world_to_image_ext(int model, const T* k, T u, T v, T& x, T& y) {
  const double eps = 2.220446049250313e-16;   // std::numeric_limits<double>::epsilon()
  switch (model) {
    case PXR_OPENCV_FISHEYE: {
      const T r = dsqrt(u * u + v * v);
      T du = T(0.0), dv = T(0.0);
      if (real_part(r) > eps) {
        const T th = datan(r), t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const T thd = th * (k[4] * t2 + k[5] * t4 + k[6] * t6 + k[7] * t8 + 1.0);
        du = u * thd / r - u; dv = v * thd / r - v;
      }
      x = k[0] * (u + du) + k[2]; y = k[1] * (v + dv) + k[3];
      return true;
    }
    case PXR_FULL_OPENCV: {
      const T u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
      const T radial = (k[4] * r2 + k[5] * r4 + k[8] * r6 + 1.0) / (k[9] * r2 + k[10] * r4 + k[11] * r6 + 1.0);
      const T du = u * radial + k[6] * uv * 2.0 + k[7] * (r2 + u2 * 2.0) - u;
      const T dv = v * radial + k[7] * uv * 2.0 + k[6] * (r2 + v2 * 2.0) - v;
      x = k[0] * (u + du) + k[2]; y = k[1] * (v + dv) + k[3];
      return true;
    }
    case PXR_FOV: {
      const T omega = k[4], radius2 = u * u + v * v, omega2 = omega * omega;
      T factor;
      if (real_part(omega2) < 1e-4) {
        // Taylor of atan(2 r tan(w/2)) / (r w) in w to 2nd order: 1 + w^2/12 - w^2 r^2/3 (continuous across the branch)
        factor = omega2 / 12.0 - (omega2 * radius2) / 3.0 + 1.0;
      } else if (real_part(radius2) < 1e-4) {
        const T tho = dtan(omega / 2.0);
        factor = (tho * -2.0 * (radius2 * 4.0 * tho * tho - 3.0)) / (omega * 3.0);
      } else {
        const T radius = dsqrt(radius2);
        factor = datan(radius * 2.0 * dtan(omega / 2.0)) / (radius * omega);
      }
      x = k[0] * (u * factor) + k[2]; y = k[1] * (v * factor) + k[3];
      return true;
    }
    case PXR_SIMPLE_RADIAL_FISHEYE:
    case PXR_RADIAL_FISHEYE: {
      const T r = dsqrt(u * u + v * v);
      T du = T(0.0), dv = T(0.0);
      if (real_part(r) > eps) {
        const T th = datan(r), t2 = th * th;
        const T rad = (model == PXR_SIMPLE_RADIAL_FISHEYE) ? k[3] * t2 : k[3] * t2 + k[4] * t2 * t2;
        const T thd = th * (rad + 1.0);
        du = u * thd / r - u; dv = v * thd / r - v;
      }
      x = k[0] * (u + du) + k[1]; y = k[0] * (v + dv) + k[2];
      return true;
    }
    case PXR_THIN_PRISM_FISHEYE: {
      T uu = u, vv = v;
      const T r = dsqrt(u * u + v * v);
      if (real_part(r) > eps) {
        const T th = datan(r);
        uu = th * u / r; vv = th * v / r;
      }
      const T u2 = uu * uu, uv = uu * vv, v2 = vv * vv, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
      const T radial = k[4] * r2 + k[5] * r4 + k[8] * r6 + k[9] * r8;
      const T du = uu * radial + k[6] * uv * 2.0 + k[7] * (r2 + u2 * 2.0) + k[10] * r2;
      const T dv = vv * radial + k[7] * uv * 2.0 + k[6] * (r2 + v2 * 2.0) + k[11] * r2;
      x = k[0] * (uu + du) + k[2]; y = k[1] * (vv + dv) + k[3];
      return true;
    }
    default:
      return false;
  }
}

// value + d(x,y)/d(u,v) (2x2) + d(x,y)/dk (2 x PXR_KPAD) by forward-mode duals
__device__ inline bool world_to_image_ext_jac(int model, const double* k, double u, double v, double& x, double& y,
                                              double Juv[2][2], double Pk[2][PXR_KPAD]) {
  constexpr int N = 2 + PXR_KPAD;
  typedef Dual<N> D;
  const int K = camera_num_params_ext(model);
  D kd[PXR_KPAD];
#pragma unroll
  for (int i = 0; i < PXR_KPAD; ++i) { kd[i] = D(k[i]); if (i < K) kd[i].v[2 + i] = 1.0; }
  D ud(u), vd(v), xd, yd;
  ud.v[0] = 1.0; vd.v[1] = 1.0;
  if (!world_to_image_ext<D>(model, kd, ud, vd, xd, yd)) return false;
  x = xd.a; y = yd.a;
  Juv[0][0] = xd.v[0]; Juv[0][1] = xd.v[1]; Juv[1][0] = yd.v[0]; Juv[1][1] = yd.v[1];
#pragma unroll
  for (int i = 0; i < PXR_KPAD; ++i) { Pk[0][i] = xd.v[2 + i]; Pk[1][i] = yd.v[2 + i]; }
  return true;
}

}  // namespace pxr

// pxr_device.h -- device-side building blocks shared by the gfx950 kernels.
//  * Catmull-Rom spline in the reference's exact operation order
//    (base/src/cubic_hermite_spline_simd.h:56-175), fp32 and fp64 flavours
//  * DPP reductions over a 16-lane row (one observation = one DPP row of a wave64)
//  * projection WorldToPixel (base/src/projection.h:60-75) with analytic Jacobians
//  * Ceres loss functions (A20)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pixsfm_hip.h"
#include "pxr_camera_ext.h"

namespace pxr {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float8_t __attribute__((ext_vector_type(8)));

// ---- spline ---------------------------------------------------------------------------
// Coefficients that depend only on the fractional offset x (formed in double, then
// rounded, exactly as the reference's _mm256_set1_ps arguments).
struct SplineCoefF32 {
  float fourx, xhalf, x2, onefivex2;
  __device__ __forceinline__ explicit SplineCoefF32(double x) {
    const float x2s = (float)(x * x);
    fourx = (float)(4.0 * x);
    xhalf = (float)(x * 0.5);
    x2 = x2s;
    onefivex2 = 1.5f * x2s;
  }
};
struct SplineCoefF64 {
  double fourx, xhalf, x2, onefivex2;
  __device__ __forceinline__ explicit SplineCoefF64(double x) {
    const double x2s = x * x;
    fourx = 4.0 * x;
    xhalf = x * 0.5;
    x2 = x2s;
    onefivex2 = 1.5 * x2s;
  }
};

// value + derivative; every fused op is explicit so the result is bit-identical to the
// AVX2 FMA sequence (fmsub(a,b,c) = fma(a,b,-c)).
template <bool WITH_D>
__device__ __forceinline__ void spline_f32(float p0, float p1, float p2, float p3,
                                           const SplineCoefF32& k, float& f, float& d) {
  const float t1 = __fmaf_rn(3.0f, p1, -p0);
  const float t2 = __fmaf_rn(3.0f, p2, -p3);
  const float t4 = __fmaf_rn(4.0f, p2, -p3);
  const float t5 = __fmaf_rn(2.5f, p1, -p0);
  const float t6 = __fmaf_rn(-1.0f, p0, p2);
  const float t3 = __fsub_rn(t1, t2);
  const float b = __fmaf_rn(0.5f, t4, -t5);
  const float t7 = __fmaf_rn(k.xhalf, t6, p1);
  const float t8 = __fmaf_rn(k.xhalf, t3, b);
  f = __fmaf_rn(k.x2, t8, t7);
  if (WITH_D) {
    const float t9 = __fmaf_rn(k.fourx, b, t6);
    const float t10 = __fmul_rn(k.onefivex2, t3);
    d = __fmaf_rn(0.5f, t9, t10);
  }
}

template <bool WITH_F, bool WITH_D>
__device__ __forceinline__ void spline_f64(double p0, double p1, double p2, double p3,
                                           const SplineCoefF64& k, double& f, double& d) {
  const double t1 = __fma_rn(3.0, p1, -p0);
  const double t2 = __fma_rn(3.0, p2, -p3);
  const double t4 = __fma_rn(4.0, p2, -p3);
  const double t5 = __fma_rn(2.5, p1, -p0);
  const double t6 = __fma_rn(-1.0, p0, p2);
  const double t3 = __dsub_rn(t1, t2);
  const double b = __fma_rn(0.5, t4, -t5);
  if (WITH_F) {
    const double t7 = __fma_rn(k.xhalf, t6, p1);
    const double t8 = __fma_rn(k.xhalf, t3, b);
    f = __fma_rn(k.x2, t8, t7);
  }
  if (WITH_D) {
    const double t9 = __fma_rn(k.fourx, b, t6);
    const double t10 = __dmul_rn(k.onefivex2, t3);
    d = __fma_rn(0.5, t9, t10);
  }
}

// ---- order-independent accumulation (deterministic mode) ---------------------------------------
// det_scale = 0: a floating-point atomic (the order of the adders decides the last bits).  det_scale = 2^k: the addend is
// rounded to a multiple of 2^-k and added as a 64-bit integer -- integer addition is associative, so the slot's final
// content does not depend on the order.  The slot then holds an integer; accum_value turns it back into a double.
__device__ __forceinline__ void accum_add(double* slot, double v, double det_scale) {
  if (det_scale == 0.0) { atomicAdd(slot, v); return; }
  atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double2ll_rn(v * det_scale));
}
// a privatised (LDS) slot's content added to its global slot: the integer as it is in deterministic mode
__device__ __forceinline__ void accum_flush(double* slot, double raw, double det_scale) {
  if (det_scale == 0.0) atomicAdd(slot, raw);
  else atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong(raw));
}
__device__ __forceinline__ double accum_value(double raw, double det_scale) {
  return det_scale == 0.0 ? raw : (double)__double_as_longlong(raw) / det_scale;
}

// ---- partition-independent scalar sums: four 40-bit limbs -----------------------------------------
// A scalar that is summed over observations / points / ranks (cost, model cost change, step norms) is accumulated as FOUR signed
// 64-bit integers: the addend's digits on the grids 2^20, 2^-20, 2^-60, 2^-100 (each digit |q| < 2^40, so 2^22 addends fit a
// limb), plus a count of addends that have no such digits (non-finite or >= 2^60).  Every addend of magnitude >= 2^-47 is
// represented exactly, the split is a function of the addend alone and integer addition is associative: the five integers --
// and the double limb_value makes of them -- do not depend on the order of the additions, on how the addends are dealt to
// threads and workgroups, or on how the observations are dealt to RANKS (the limbs are all-reduced as integers).
constexpr int PXR_LIMBS = 5;      // four digits + the count of unrepresentable addends
struct Limbs {
  long long q[PXR_LIMBS];
  __device__ __forceinline__ Limbs() { for (int k = 0; k < PXR_LIMBS; ++k) q[k] = 0; }
  __device__ __forceinline__ void add(double x) {
    if (!(fabs(x) < 0x1p60)) { ++q[4]; return; }                       // NaN, Inf, out of range
    double t = trunc(x * 0x1p-20); q[0] += (long long)t; x -= t * 0x1p20;      // every step exact: the remainder keeps x's sign
    t = trunc(x * 0x1p20); q[1] += (long long)t; x -= t * 0x1p-20;
    t = trunc(x * 0x1p60); q[2] += (long long)t; x -= t * 0x1p-60;
    q[3] += (long long)rint(x * 0x1p100);
  }
};
__device__ __forceinline__ long long wave_sum_ll(long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// the workgroup's limbs added to the global slot (PXR_LIMBS integer atomics per workgroup); `sh`: PXR_LIMBS * (blockDim / 64) slots of LDS
__device__ __forceinline__ void limbs_block_add(Limbs l, long long* __restrict__ slot, long long* sh) {
  const int nw = (int)blockDim.x >> 6, w = (int)threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < PXR_LIMBS; ++k) {
    const long long v = wave_sum_ll(l.q[k]);
    if ((threadIdx.x & 63) == 0) sh[k * nw + w] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < PXR_LIMBS) {
    long long v = 0;
    for (int i = 0; i < nw; ++i) v += sh[threadIdx.x * nw + i];
    if (v != 0) atomicAdd(reinterpret_cast<unsigned long long*>(slot + threadIdx.x), (unsigned long long)v);
  }
}
__device__ __forceinline__ double limb_value(const long long* q) {
  if (q[4] != 0) return __longlong_as_double(0x7ff8000000000000ll);   // an addend could not be represented: NaN (rejects the step)
  return (((double)q[3] * 0x1p-100 + (double)q[2] * 0x1p-60) + (double)q[1] * 0x1p-20) + (double)q[0] * 0x1p20;
}

// ---- 16-lane (DPP row) all-reduce --------------------------------------------------------
// quad_perm[1,0,3,2] -> quad_perm[2,3,0,1] -> row_half_mirror -> row_mirror: after the four
// steps every lane of the row holds the row sum.  No LDS, no ds_bpermute.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  union { double d; int i[2]; } a, b;
  a.d = v;
  b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, 0xf, 0xf, true);
  b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, 0xf, 0xf, true);
  return b.d;
}
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return v;
}
// 8-lane variant (C = 64: 8 lanes per observation, two observations per DPP row)
__device__ __forceinline__ double row8_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  return v;
}

// ---- camera models [upstream COLMAP 3.8 WorldToImage], values only ------------------------
__device__ __forceinline__ bool world_to_image(int model, const double* __restrict__ k, double u,
                                               double v, double& x, double& y) {
  const double u2 = u * u, v2 = v * v, uv = u * v, r2 = u2 + v2;
  switch (model) {
    case PXR_SIMPLE_PINHOLE:
      x = k[0] * u + k[1]; y = k[0] * v + k[2];
      return true;
    case PXR_PINHOLE:
      x = k[0] * u + k[2]; y = k[1] * v + k[3];
      return true;
    case PXR_SIMPLE_RADIAL: {
      const double radial = k[3] * r2;
      x = k[0] * (u + u * radial) + k[1]; y = k[0] * (v + v * radial) + k[2];
      return true;
    }
    case PXR_RADIAL: {
      const double radial = k[3] * r2 + k[4] * r2 * r2;
      x = k[0] * (u + u * radial) + k[1]; y = k[0] * (v + v * radial) + k[2];
      return true;
    }
    case PXR_OPENCV: {
      const double radial = k[4] * r2 + k[5] * r2 * r2;
      const double du = u * radial + 2.0 * k[6] * uv + k[7] * (r2 + 2.0 * u2);
      const double dv = v * radial + 2.0 * k[7] * uv + k[6] * (r2 + 2.0 * v2);
      x = k[0] * (u + du) + k[2]; y = k[1] * (v + dv) + k[3];
      return true;
    }
    default:   // OPENCV_FISHEYE ... THIN_PRISM_FISHEYE (pxr_camera_ext.h)
      if (world_to_image_ext<double>(model, k, u, v, x, y)) return true;
      x = y = 0.0;
      return false;
  }
}

__device__ __forceinline__ int camera_num_params(int model) {
  switch (model) {
    case PXR_SIMPLE_PINHOLE: return 3;
    case PXR_PINHOLE: return 4;
    case PXR_SIMPLE_RADIAL: return 4;
    case PXR_RADIAL: return 5;
    case PXR_OPENCV: return 8;
    default: return camera_num_params_ext(model);
  }
}

// [upstream Ceres 2.1 rotation.h] QuaternionRotatePoint (normalises q) + translation.
__device__ __forceinline__ void rotate_translate(const double* __restrict__ q,
                                                 const double* __restrict__ t,
                                                 const double* __restrict__ X, double p[3]) {
  const double scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * scale, a = q[1] * scale, b = q[2] * scale, c = q[3] * scale;
  double uv0 = b * X[2] - c * X[1];
  double uv1 = c * X[0] - a * X[2];
  double uv2 = a * X[1] - b * X[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  p[0] = X[0] + w * uv0 + (b * uv2 - c * uv1) + t[0];
  p[1] = X[1] + w * uv1 + (c * uv0 - a * uv2) + t[1];
  p[2] = X[2] + w * uv2 + (a * uv1 - b * uv0) + t[2];
}

// Rotation matrix (row-major) of the NORMALISED quaternion q = (w, x, y, z): the dp/dX block of world_to_pixel_jac.
__device__ __forceinline__ void quat_to_rotation(const double* __restrict__ q, double R[9]) {
  const double scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * scale, a = q[1] * scale, b = q[2] * scale, c = q[3] * scale;
  R[0] = 1.0 - 2.0 * (b * b + c * c); R[1] = 2.0 * (a * b - w * c); R[2] = 2.0 * (a * c + w * b);
  R[3] = 2.0 * (a * b + w * c); R[4] = 1.0 - 2.0 * (a * a + c * c); R[5] = 2.0 * (b * c - w * a);
  R[6] = 2.0 * (a * c - w * b); R[7] = 2.0 * (b * c + w * a); R[8] = 1.0 - 2.0 * (a * a + b * b);
}

// WorldToPixel value (base/src/projection.h:60-75).
__device__ __forceinline__ bool world_to_pixel(int model, const double* __restrict__ k,
                                               const double* __restrict__ q,
                                               const double* __restrict__ t,
                                               const double* __restrict__ X, double& x, double& y) {
  double p[3];
  rotate_translate(q, t, X, p);
  return world_to_image(model, k, p[0] / p[2], p[1] / p[2], x, y);
}

// Camera model at the normalised image point (u, v): value, d(x,y)/d(u,v) (Juv) and -- WITH_PK -- d(x,y)/dk (Pk,
// PXR_KPAD-strided rows; not touched otherwise: the callers that only move the point pass nullptr).
// WITH_EXT = false: without the six fisheye / full-OpenCV / FOV models (forward-mode duals: ~100 registers) -- for kernels whose
// host routes problems with those models elsewhere; such a model then fails the evaluation.
template <bool WITH_PK = true, bool WITH_EXT = true>
__device__ __forceinline__ bool camera_model_jac(int model, const double* __restrict__ k, double u, double v, double& x,
                                                 double& y, double Juv[2][2], double Pk[2][PXR_KPAD]) {
  const double u2 = u * u, v2 = v * v, uvp = u * v, r2 = u2 + v2;
  double fx, fy, cx, cy, du = 0, dv = 0, duu = 0, duv = 0, dvu = 0, dvv = 0;
  if constexpr (WITH_PK) {
#pragma unroll
    for (int j = 0; j < PXR_KPAD; ++j) { Pk[0][j] = 0.0; Pk[1][j] = 0.0; }
  }
  bool two_focal = false, ext_model = false;
  switch (model) {
    case PXR_SIMPLE_PINHOLE:
      fx = fy = k[0]; cx = k[1]; cy = k[2];
      break;
    case PXR_PINHOLE:
      fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; two_focal = true;
      break;
    case PXR_SIMPLE_RADIAL: {
      fx = fy = k[0]; cx = k[1]; cy = k[2];
      const double kk = k[3], radial = kk * r2;
      du = u * radial; dv = v * radial;
      duu = radial + 2.0 * kk * u2; duv = 2.0 * kk * uvp; dvu = duv; dvv = radial + 2.0 * kk * v2;
      if constexpr (WITH_PK) { Pk[0][3] = fx * u * r2; Pk[1][3] = fy * v * r2; }
      break;
    }
    case PXR_RADIAL: {
      fx = fy = k[0]; cx = k[1]; cy = k[2];
      const double radial = k[3] * r2 + k[4] * r2 * r2, rp = 2.0 * k[3] + 4.0 * k[4] * r2;
      du = u * radial; dv = v * radial;
      duu = radial + rp * u2; duv = rp * uvp; dvu = duv; dvv = radial + rp * v2;
      if constexpr (WITH_PK) {
        Pk[0][3] = fx * u * r2; Pk[0][4] = fx * u * r2 * r2;
        Pk[1][3] = fy * v * r2; Pk[1][4] = fy * v * r2 * r2;
      }
      break;
    }
    case PXR_OPENCV: {
      fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3]; two_focal = true;
      const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7];
      const double radial = k1 * r2 + k2 * r2 * r2, rp = 2.0 * k1 + 4.0 * k2 * r2;
      du = u * radial + 2.0 * p1 * uvp + p2 * (r2 + 2.0 * u2);
      dv = v * radial + 2.0 * p2 * uvp + p1 * (r2 + 2.0 * v2);
      duu = radial + rp * u2 + 2.0 * p1 * v + 6.0 * p2 * u;
      duv = rp * uvp + 2.0 * p1 * u + 2.0 * p2 * v;
      dvu = rp * uvp + 2.0 * p2 * v + 2.0 * p1 * u;
      dvv = radial + rp * v2 + 2.0 * p2 * u + 6.0 * p1 * v;
      if constexpr (WITH_PK) {
        Pk[0][4] = fx * u * r2; Pk[0][5] = fx * u * r2 * r2; Pk[0][6] = fx * 2.0 * uvp;
        Pk[0][7] = fx * (r2 + 2.0 * u2);
        Pk[1][4] = fy * v * r2; Pk[1][5] = fy * v * r2 * r2; Pk[1][6] = fy * (r2 + 2.0 * v2);
        Pk[1][7] = fy * 2.0 * uvp;
      }
      break;
    }
    default:
      ext_model = true;
      fx = fy = cx = cy = 0.0;
      break;
  }
  if (ext_model) {   // forward-mode duals (pxr_camera_ext.h)
    if constexpr (WITH_EXT) {
      double Pk_ext[2][PXR_KPAD];
      if (!world_to_image_ext_jac(model, k, u, v, x, y, Juv, WITH_PK ? Pk : Pk_ext)) { x = y = 0.0; return false; }
    } else {
      x = y = 0.0; Juv[0][0] = Juv[0][1] = Juv[1][0] = Juv[1][1] = 0.0;
      return false;
    }
  } else {
    x = fx * (u + du) + cx;
    y = fy * (v + dv) + cy;
    if constexpr (WITH_PK) {
      if (two_focal) {
        Pk[0][0] = u + du; Pk[1][1] = v + dv; Pk[0][2] = 1.0; Pk[1][3] = 1.0;
      } else {
        Pk[0][0] = u + du; Pk[1][0] = v + dv; Pk[0][1] = 1.0; Pk[1][2] = 1.0;
      }
    }
    Juv[0][0] = fx * (1.0 + duu); Juv[0][1] = fx * duv; Juv[1][0] = fy * dvu; Juv[1][1] = fy * (1.0 + dvv);
  }
  return true;
}

// WorldToPixel with analytic Jacobians.  A = d(x,y)/dp (2x3); Pq (2x4, ambient, includes
// the normalisation Jacobian), PX (2x3), Pk (2xK in PXR_KPAD-strided rows).
__device__ inline bool world_to_pixel_jac(int model, const double* __restrict__ k,
                                          const double* __restrict__ q,
                                          const double* __restrict__ t,
                                          const double* __restrict__ X, double& x, double& y,
                                          double A[2][3], double Pq[2][4], double PX[2][3],
                                          double Pk[2][PXR_KPAD]) {
  const double scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * scale, a = q[1] * scale, b = q[2] * scale, c = q[3] * scale;
  const double cr0 = b * X[2] - c * X[1], cr1 = c * X[0] - a * X[2], cr2 = a * X[1] - b * X[0];
  const double uv0 = 2.0 * cr0, uv1 = 2.0 * cr1, uv2 = 2.0 * cr2;
  double p[3];
  p[0] = X[0] + w * uv0 + (b * uv2 - c * uv1) + t[0];
  p[1] = X[1] + w * uv1 + (c * uv0 - a * uv2) + t[1];
  p[2] = X[2] + w * uv2 + (a * uv1 - b * uv0) + t[2];
  const double iz = 1.0 / p[2];
  const double u = p[0] * iz, v = p[1] * iz;
  double Juv[2][2];
  if (!camera_model_jac(model, k, u, v, x, y, Juv, Pk)) return false;
  const double D[2][3] = {{iz, 0.0, -p[0] * iz * iz}, {0.0, iz, -p[1] * iz * iz}};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) A[i][j] = Juv[i][0] * D[0][j] + Juv[i][1] * D[1][j];
  // dp/dX = R(unit quaternion)
  const double R[3][3] = {
      {1.0 - 2.0 * (b * b + c * c), 2.0 * (a * b - w * c), 2.0 * (a * c + w * b)},
      {2.0 * (a * b + w * c), 1.0 - 2.0 * (a * a + c * c), 2.0 * (b * c - w * a)},
      {2.0 * (a * c - w * b), 2.0 * (b * c + w * a), 1.0 - 2.0 * (a * a + b * b)}};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      PX[i][j] = A[i][0] * R[0][j] + A[i][1] * R[1][j] + A[i][2] * R[2][j];
  // dp/d(unit) then d(unit)/dq = scale (I - unit unit^T)
  const double vv[3] = {a, b, c};
  const double vX = a * X[0] + b * X[1] + c * X[2];
  const double cr[3] = {cr0, cr1, cr2};
  const double Xx[3][3] = {{0.0, -X[2], X[1]}, {X[2], 0.0, -X[0]}, {-X[1], X[0], 0.0}};
  const double un4[4] = {w, a, b, c};
  double Pqq[3][4];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double Pu[4];
    Pu[0] = 2.0 * cr[i];
#pragma unroll
    for (int j = 0; j < 3; ++j)
      Pu[1 + j] = -2.0 * w * Xx[i][j] + 2.0 * ((i == j ? vX : 0.0) + vv[i] * X[j] - 2.0 * X[i] * vv[j]);
    const double dotu = Pu[0] * un4[0] + Pu[1] * un4[1] + Pu[2] * un4[2] + Pu[3] * un4[3];
#pragma unroll
    for (int j = 0; j < 4; ++j) Pqq[i][j] = scale * (Pu[j] - dotu * un4[j]);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      Pq[i][j] = A[i][0] * Pqq[0][j] + A[i][1] * Pqq[1][j] + A[i][2] * Pqq[2][j];
  return true;
}

// ---- robustifier [upstream Ceres 2.1 loss_function.cc] --------------------------------------
__device__ __forceinline__ void loss_eval(int type, double a, double weight, double s,
                                          double rho[3]) {
  const double b = a * a;
  const double tiny = 2.2250738585072014e-308;
  switch (type) {
    case PXR_LOSS_CAUCHY: {
      const double c = 1.0 / b, sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * log(sum); rho[1] = fmax(tiny, inv); rho[2] = -c * (inv * inv);
      break;
    }
    case PXR_LOSS_HUBER:
      if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b; rho[1] = fmax(tiny, a / r); rho[2] = -rho[1] / (2.0 * s);
      } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
      }
      break;
    case PXR_LOSS_SOFTL1: {
      const double c = 1.0 / b, sum = 1.0 + s * c, tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0); rho[1] = fmax(tiny, 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      break;
    }
    default:
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
  rho[0] *= weight; rho[1] *= weight; rho[2] *= weight;
}

}  // namespace pxr

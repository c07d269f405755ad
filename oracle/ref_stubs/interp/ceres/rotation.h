// Stub for <ceres/rotation.h>: the two rotation helpers base/src/projection.h calls, restated from their published
// definitions ([upstream Ceres 2.x] rotation.h: UnitQuaternionRotatePoint's expanded form, QuaternionRotatePoint =
// normalise + that; quaternion order w, x, y, z).  NOT reference code: parity of a projection rests on this stub as far
// as the rotation itself goes.
#pragma once
#include "ceres/ceres.h"
namespace ceres {
template <typename T>
inline void UnitQuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  T uv0 = q[2] * pt[2] - q[3] * pt[1];
  T uv1 = q[3] * pt[0] - q[1] * pt[2];
  T uv2 = q[1] * pt[1] - q[2] * pt[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  result[0] = pt[0] + q[0] * uv0;
  result[1] = pt[1] + q[0] * uv1;
  result[2] = pt[2] + q[0] * uv2;
  result[0] += q[2] * uv2 - q[3] * uv1;
  result[1] += q[3] * uv0 - q[1] * uv2;
  result[2] += q[1] * uv1 - q[2] * uv0;
}
template <typename T>
inline void QuaternionRotatePoint(const T q[4], const T pt[3], T result[3]) {
  const T scale = T(1) / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const T unit[4] = {scale * q[0], scale * q[1], scale * q[2], scale * q[3]};
  UnitQuaternionRotatePoint(unit, pt, result);
}
template <typename T>
inline void QuaternionToRotation(const T q[4], T R[9]) {
  const T a = q[0], b = q[1], c = q[2], d = q[3];
  const T aa = a * a, ab = a * b, ac = a * c, ad = a * d, bb = b * b, bc = b * c, bd = b * d, cc = c * c, cd = c * d, dd = d * d;
  R[0] = aa + bb - cc - dd; R[1] = T(2) * (bc - ad); R[2] = T(2) * (ac + bd);
  R[3] = T(2) * (ad + bc); R[4] = aa - bb + cc - dd; R[5] = T(2) * (cd - ab);
  R[6] = T(2) * (bd - ac); R[7] = T(2) * (ab + cd); R[8] = aa - bb - cc + dd;
}
}  // namespace ceres

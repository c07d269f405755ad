/*
 * pxo_camera_ext.c -- ORACLE (test infrastructure only; see pxo.h header).
 * The remaining COLMAP 3.8 camera models of CAMERA_MODEL_SWITCH_CASES
 * (residuals/src/feature_reference.h:232 -> [upstream colmap/base/camera_models.h]):
 * OPENCV_FISHEYE (5), FULL_OPENCV (6), FOV (7), SIMPLE_RADIAL_FISHEYE (8), RADIAL_FISHEYE (9),
 * THIN_PRISM_FISHEYE (10).  WorldToImage is restated from the published COLMAP formulas in
 * complex arithmetic, and the Jacobians the reference obtains by ceres::Jet autodiff are obtained
 * here by COMPLEX-STEP differentiation (Im f(x + ih) / h, h = 1e-30: exact to rounding, no
 * subtractive cancellation) -- an independent derivation from the hand-written analytic
 * Jacobians of the HIP kernels.
 */
#include <complex.h>
#include <math.h>

#include "pxo.h"

typedef double complex cx;

static const double kEps = 2.220446049250313e-16; /* std::numeric_limits<double>::epsilon() */

static void fisheye_theta(cx u, cx v, cx* uu, cx* vv) { /* r > eps: theta / r scaling */
  const cx r = csqrt(u * u + v * v);
  if (creal(r) > kEps) {
    const cx theta = catan(r);
    *uu = theta * u / r; *vv = theta * v / r;
  } else {
    *uu = u; *vv = v;
  }
}

/* returns 0 on success; x, y complex so that any input may carry the imaginary perturbation */
static int w2i_cx(int model, const cx* k, cx u, cx v, cx* x, cx* y) {
  switch (model) {
    case PXO_OPENCV_FISHEYE: { /* fx, fy, cx, cy, k1, k2, k3, k4 */
      const cx r = csqrt(u * u + v * v);
      cx du = 0, dv = 0;
      if (creal(r) > kEps) {
        const cx th = catan(r), t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
        const cx thd = th * (1.0 + k[4] * t2 + k[5] * t4 + k[6] * t6 + k[7] * t8);
        du = u * thd / r - u; dv = v * thd / r - v;
      }
      *x = k[0] * (u + du) + k[2]; *y = k[1] * (v + dv) + k[3];
      return 0;
    }
    case PXO_FULL_OPENCV: { /* fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, k5, k6 */
      const cx u2 = u * u, uv = u * v, v2 = v * v, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2;
      const cx radial = (1.0 + k[4] * r2 + k[5] * r4 + k[8] * r6) / (1.0 + k[9] * r2 + k[10] * r4 + k[11] * r6);
      const cx du = u * radial + 2.0 * k[6] * uv + k[7] * (r2 + 2.0 * u2) - u;
      const cx dv = v * radial + 2.0 * k[7] * uv + k[6] * (r2 + 2.0 * v2) - v;
      *x = k[0] * (u + du) + k[2]; *y = k[1] * (v + dv) + k[3];
      return 0;
    }
    case PXO_FOV: { /* fx, fy, cx, cy, omega; Distortion returns the distorted coordinates */
      const cx omega = k[4], radius2 = u * u + v * v, omega2 = omega * omega;
      cx factor;
      if (creal(omega2) < 1e-4) {
        /* Taylor of atan(2 r tan(w/2)) / (r w) in w to 2nd order: 1 + w^2/12 - w^2 r^2/3.  [upstream: COLMAP's
         * comment names this expansion; its literal coefficients cannot be checked offline.] */
        factor = omega2 / 12.0 - (omega2 * radius2) / 3.0 + 1.0;
      } else if (creal(radius2) < 1e-4) {
        const cx tho = ctan(omega / 2.0);
        factor = (-2.0 * tho * (4.0 * radius2 * tho * tho - 3.0)) / (3.0 * omega);
      } else {
        const cx radius = csqrt(radius2);
        factor = catan(radius * 2.0 * ctan(omega / 2.0)) / (radius * omega);
      }
      *x = k[0] * (u * factor) + k[2]; *y = k[1] * (v * factor) + k[3];
      return 0;
    }
    case PXO_SIMPLE_RADIAL_FISHEYE:   /* f, cx, cy, k */
    case PXO_RADIAL_FISHEYE: {        /* f, cx, cy, k1, k2 */
      const cx r = csqrt(u * u + v * v);
      cx du = 0, dv = 0;
      if (creal(r) > kEps) {
        const cx th = catan(r), t2 = th * th;
        const cx rad = (model == PXO_SIMPLE_RADIAL_FISHEYE) ? k[3] * t2 : k[3] * t2 + k[4] * t2 * t2;
        const cx thd = th * (1.0 + rad);
        du = u * thd / r - u; dv = v * thd / r - v;
      }
      *x = k[0] * (u + du) + k[1]; *y = k[0] * (v + dv) + k[2];
      return 0;
    }
    case PXO_THIN_PRISM_FISHEYE: { /* fx, fy, cx, cy, k1, k2, p1, p2, k3, k4, sx1, sy1 */
      cx uu, vv;
      fisheye_theta(u, v, &uu, &vv);
      const cx u2 = uu * uu, uv = uu * vv, v2 = vv * vv, r2 = u2 + v2, r4 = r2 * r2, r6 = r4 * r2, r8 = r4 * r4;
      const cx radial = k[4] * r2 + k[5] * r4 + k[8] * r6 + k[9] * r8;
      const cx du = uu * radial + 2.0 * k[6] * uv + k[7] * (r2 + 2.0 * u2) + k[10] * r2;
      const cx dv = vv * radial + 2.0 * k[7] * uv + k[6] * (r2 + 2.0 * v2) + k[11] * r2;
      *x = k[0] * (uu + du) + k[2]; *y = k[1] * (vv + dv) + k[3];
      return 0;
    }
    default:
      return -1;
  }
}

int pxo_camera_num_params_ext(int model) {
  switch (model) {
    case PXO_OPENCV_FISHEYE: return 8;
    case PXO_FULL_OPENCV: return 12;
    case PXO_FOV: return 5;
    case PXO_SIMPLE_RADIAL_FISHEYE: return 4;
    case PXO_RADIAL_FISHEYE: return 5;
    case PXO_THIN_PRISM_FISHEYE: return 12;
    default: return -1;
  }
}

int pxo_world_to_image_ext(int model, const double* k, double u, double v, double* x, double* y,
                           double* J_uv, double* J_k) {
  const int K = pxo_camera_num_params_ext(model);
  if (K < 0) return -1;
  cx kc[12], xx, yy;
  for (int i = 0; i < K; ++i) kc[i] = k[i];
  if (w2i_cx(model, kc, u, v, &xx, &yy)) return -1;
  *x = creal(xx); *y = creal(yy);
  const double h = 1e-30;
  if (J_uv) {
    w2i_cx(model, kc, u + h * I, v, &xx, &yy); J_uv[0] = cimag(xx) / h; J_uv[2] = cimag(yy) / h;
    w2i_cx(model, kc, u, v + h * I, &xx, &yy); J_uv[1] = cimag(xx) / h; J_uv[3] = cimag(yy) / h;
  }
  if (J_k) {
    for (int i = 0; i < K; ++i) {
      kc[i] = k[i] + h * I;
      w2i_cx(model, kc, u, v, &xx, &yy);
      J_k[0 * K + i] = cimag(xx) / h; J_k[1 * K + i] = cimag(yy) / h;
      kc[i] = k[i];
    }
  }
  return 0;
}

/*
 * pxo_geom.c -- ORACLE (test infrastructure only; see pxo.h header).
 * Restates A6-A10, A19, A20 of SURVEY.md section 8a: projection with hand-derived
 * analytic Jacobians (the reference obtains them by ceres::Jet autodiff), the residual
 * functors, Ceres' loss functions + corrector, and the IRLS reference extraction.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "pxo.h"

#define PXO_MAXC 512
#define PXO_KPAD 12

int pxo_camera_num_params_ext(int model);
int pxo_world_to_image_ext(int model, const double* k, double u, double v, double* x, double* y,
                           double* J_uv, double* J_k);

/* [upstream COLMAP 3.8 camera_models.h] kNumParams */
int pxo_camera_num_params(int model) {
  switch (model) {
    case PXO_SIMPLE_PINHOLE: return 3;
    case PXO_PINHOLE: return 4;
    case PXO_SIMPLE_RADIAL: return 4;
    case PXO_RADIAL: return 5;
    case PXO_OPENCV: return 8;
    default: return pxo_camera_num_params_ext(model);   /* pxo_camera_ext.c */
  }
}

/* [upstream COLMAP 3.8] CameraModel::WorldToImage, called at base/src/projection.h:73.
 *   SIMPLE_PINHOLE (f,cx,cy)             x = f u + cx
 *   PINHOLE (fx,fy,cx,cy)
 *   SIMPLE_RADIAL (f,cx,cy,k)            du = u k r2
 *   RADIAL (f,cx,cy,k1,k2)               du = u (k1 r2 + k2 r2^2)
 *   OPENCV (fx,fy,cx,cy,k1,k2,p1,p2)     du = u radial + 2 p1 uv + p2 (r2 + 2u^2)
 *                                        dv = v radial + 2 p2 uv + p1 (r2 + 2v^2)      */
int pxo_world_to_image(int model, const double* k, double u, double v, double* x, double* y,
                       double* J_uv, double* J_k) {
  const int K = pxo_camera_num_params(model);
  if (K < 0) return -1;
  if (model > PXO_OPENCV) return pxo_world_to_image_ext(model, k, u, v, x, y, J_uv, J_k);
  double fx, fy, cx, cy;
  double du = 0, dv = 0, duu = 0, duv = 0, dvu = 0, dvv = 0; /* d(du)/du ... */
  const double u2 = u * u, v2 = v * v, uv = u * v, r2 = u2 + v2;
  if (J_k) memset(J_k, 0, sizeof(double) * 2 * K);
  switch (model) {
    case PXO_SIMPLE_PINHOLE:
      fx = fy = k[0]; cx = k[1]; cy = k[2];
      break;
    case PXO_PINHOLE:
      fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3];
      break;
    case PXO_SIMPLE_RADIAL: {
      fx = fy = k[0]; cx = k[1]; cy = k[2];
      const double kk = k[3];
      const double radial = kk * r2;
      du = u * radial; dv = v * radial;
      duu = radial + 2 * kk * u2; duv = 2 * kk * uv;
      dvu = 2 * kk * uv;          dvv = radial + 2 * kk * v2;
      if (J_k) { J_k[0 * K + 3] = fx * u * r2; J_k[1 * K + 3] = fy * v * r2; }
      break;
    }
    case PXO_RADIAL: {
      fx = fy = k[0]; cx = k[1]; cy = k[2];
      const double k1 = k[3], k2 = k[4];
      const double radial = k1 * r2 + k2 * r2 * r2;
      const double rp = 2 * k1 + 4 * k2 * r2; /* d radial / du = rp*u */
      du = u * radial; dv = v * radial;
      duu = radial + rp * u2; duv = rp * uv;
      dvu = rp * uv;          dvv = radial + rp * v2;
      if (J_k) {
        J_k[0 * K + 3] = fx * u * r2; J_k[0 * K + 4] = fx * u * r2 * r2;
        J_k[1 * K + 3] = fy * v * r2; J_k[1 * K + 4] = fy * v * r2 * r2;
      }
      break;
    }
    case PXO_OPENCV: {
      fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3];
      const double k1 = k[4], k2 = k[5], p1 = k[6], p2 = k[7];
      const double radial = k1 * r2 + k2 * r2 * r2;
      const double rp = 2 * k1 + 4 * k2 * r2;
      du = u * radial + 2 * p1 * uv + p2 * (r2 + 2 * u2);
      dv = v * radial + 2 * p2 * uv + p1 * (r2 + 2 * v2);
      duu = radial + rp * u2 + 2 * p1 * v + 6 * p2 * u;
      duv = rp * uv + 2 * p1 * u + 2 * p2 * v;
      dvu = rp * uv + 2 * p2 * v + 2 * p1 * u;
      dvv = radial + rp * v2 + 2 * p2 * u + 6 * p1 * v;
      if (J_k) {
        J_k[0 * K + 4] = fx * u * r2; J_k[0 * K + 5] = fx * u * r2 * r2;
        J_k[0 * K + 6] = fx * 2 * uv; J_k[0 * K + 7] = fx * (r2 + 2 * u2);
        J_k[1 * K + 4] = fy * v * r2; J_k[1 * K + 5] = fy * v * r2 * r2;
        J_k[1 * K + 6] = fy * (r2 + 2 * v2); J_k[1 * K + 7] = fy * 2 * uv;
      }
      break;
    }
    default:
      return -1;
  }
  *x = fx * (u + du) + cx;
  *y = fy * (v + dv) + cy;
  if (J_uv) {
    J_uv[0] = fx * (1 + duu); J_uv[1] = fx * duv;
    J_uv[2] = fy * dvu;       J_uv[3] = fy * (1 + dvv);
  }
  if (J_k) {
    if (model == PXO_PINHOLE || model == PXO_OPENCV) {
      J_k[0 * K + 0] = u + du; J_k[1 * K + 1] = v + dv;
      J_k[0 * K + 2] = 1;      J_k[1 * K + 3] = 1;
    } else {
      J_k[0 * K + 0] = u + du; J_k[1 * K + 0] = v + dv;
      J_k[0 * K + 1] = 1;      J_k[1 * K + 2] = 1;
    }
  }
  return 0;
}

/* WorldToPixel (base/src/projection.h:60-75) with
 * [upstream Ceres 2.1 rotation.h] QuaternionRotatePoint: unit = q/|q|, then
 * UnitQuaternionRotatePoint: uv = 2 (qv x X); p = X + w uv + qv x uv.              */
int pxo_world_to_pixel(int model, const double* params, const double q[4], const double t[3],
                       const double X[3], double xy[2], double* J_q, double* J_t, double* J_X,
                       double* J_k) {
  const double scale = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double w = q[0] * scale, a = q[1] * scale, b = q[2] * scale, c = q[3] * scale;
  double uv0 = b * X[2] - c * X[1];
  double uv1 = c * X[0] - a * X[2];
  double uv2 = a * X[1] - b * X[0];
  uv0 += uv0; uv1 += uv1; uv2 += uv2;
  double p[3];
  p[0] = X[0] + w * uv0; p[1] = X[1] + w * uv1; p[2] = X[2] + w * uv2;
  p[0] += b * uv2 - c * uv1;
  p[1] += c * uv0 - a * uv2;
  p[2] += a * uv1 - b * uv0;
  p[0] += t[0]; p[1] += t[1]; p[2] += t[2];
  const double un = p[0] / p[2], vn = p[1] / p[2];
  double Juv[4];
  const int K = pxo_camera_num_params(model);
  if (K < 0) return -1;
  if (pxo_world_to_image(model, params, un, vn, &xy[0], &xy[1], Juv, J_k)) return -1;
  if (!J_q && !J_t && !J_X) return 0;
  /* d(un,vn)/dp */
  const double iz = 1.0 / p[2];
  const double D[2][3] = {{iz, 0, -p[0] * iz * iz}, {0, iz, -p[1] * iz * iz}};
  double A[2][3]; /* d(x,y)/dp = Juv * D */
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) A[i][j] = Juv[i * 2 + 0] * D[0][j] + Juv[i * 2 + 1] * D[1][j];
  if (J_t)
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) J_t[i * 3 + j] = A[i][j];
  if (J_X) { /* dp/dX = R(unit) */
    const double R[3][3] = {
        {1 - 2 * (b * b + c * c), 2 * (a * b - w * c), 2 * (a * c + w * b)},
        {2 * (a * b + w * c), 1 - 2 * (a * a + c * c), 2 * (b * c - w * a)},
        {2 * (a * c - w * b), 2 * (b * c + w * a), 1 - 2 * (a * a + b * b)}};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j)
        J_X[i * 3 + j] = A[i][0] * R[0][j] + A[i][1] * R[1][j] + A[i][2] * R[2][j];
  }
  if (J_q) {
    /* dp/d(unit): p = X + 2w (v x X) + 2 v x (v x X), v = (a,b,c)
     *   d/dw = 2 (v x X)
     *   d/dv = -2w [X]x + 2 ((v.X) I + v X^T - 2 X v^T)                               */
    const double v[3] = {a, b, c};
    const double vX = a * X[0] + b * X[1] + c * X[2];
    const double cr[3] = {b * X[2] - c * X[1], c * X[0] - a * X[2], a * X[1] - b * X[0]};
    const double Xx[3][3] = {{0, -X[2], X[1]}, {X[2], 0, -X[0]}, {-X[1], X[0], 0}};
    double Pu[3][4];
    for (int i = 0; i < 3; ++i) {
      Pu[i][0] = 2 * cr[i];
      for (int j = 0; j < 3; ++j)
        Pu[i][1 + j] = -2 * w * Xx[i][j] + 2 * ((i == j ? vX : 0.0) + v[i] * X[j] - 2 * X[i] * v[j]);
    }
    /* d(unit)/dq = scale (I - unit unit^T) */
    const double un4[4] = {w, a, b, c};
    double Pq[3][4];
    for (int i = 0; i < 3; ++i) {
      double dotu = 0;
      for (int m = 0; m < 4; ++m) dotu += Pu[i][m] * un4[m];
      for (int j = 0; j < 4; ++j) Pq[i][j] = scale * (Pu[i][j] - dotu * un4[j]);
    }
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 4; ++j)
        J_q[i * 4 + j] = A[i][0] * Pq[0][j] + A[i][1] * Pq[1][j] + A[i][2] * Pq[2][j];
  }
  return 0;
}

/* FeatureReferenceCostFunctor::operator() (residuals/src/feature_reference.h:98-137),
 * N_NODES = 1 branch :121-124, reference subtraction :132-134.  Jacobian = G*P
 * (Jet bridge base/src/interpolation.h:130-140).                                      */
int pxo_ba_residual(const pxo_patch* p, const pxo_interp_cfg* cfg, int model, const double q[4],
                    const double t[3], const double X[3], const double* params,
                    const double* ref, double* r, double* Jq, double* Jt, double* JX,
                    double* Jk) {
  const int C = p->C, K = pxo_camera_num_params(model);
  if (K < 0 || C > PXO_MAXC) return -1;
  double xy[2], Pq[8], Pt[6], PX[6], Pk[2 * PXO_KPAD];
  const int wantJ = Jq || Jt || JX || Jk;
  if (pxo_world_to_pixel(model, params, q, t, X, xy, wantJ ? Pq : NULL, wantJ ? Pt : NULL,
                         wantJ ? PX : NULL, wantJ ? Pk : NULL))
    return -1;
  double gx[PXO_MAXC], gy[PXO_MAXC];
  int inside = pxo_patch_eval(p, xy, cfg, r, wantJ ? gx : NULL, wantJ ? gy : NULL);
  if (ref)
    for (int i = 0; i < C; ++i) r[i] -= ref[i];
  for (int i = 0; i < C && wantJ; ++i) {
    if (Jq) for (int j = 0; j < 4; ++j) Jq[i * 4 + j] = gx[i] * Pq[j] + gy[i] * Pq[4 + j];
    if (Jt) for (int j = 0; j < 3; ++j) Jt[i * 3 + j] = gx[i] * Pt[j] + gy[i] * Pt[3 + j];
    if (JX) for (int j = 0; j < 3; ++j) JX[i * 3 + j] = gx[i] * PX[j] + gy[i] * PX[3 + j];
    if (Jk) for (int j = 0; j < K; ++j) Jk[i * K + j] = gx[i] * Pk[j] + gy[i] * Pk[K + j];
  }
  return ref ? 1 : inside; /* feature_reference.h:128-136 */
}

/* FeatureMetric2DCostFunctor::operator() (residuals/src/featuremetric.h:44-63). */
int pxo_ka_residual(const pxo_patch* p1, const pxo_patch* p2, const pxo_interp_cfg* cfg,
                    const double kp1[2], const double kp2[2], double* r, double* J1,
                    double* J2) {
  const int C = p1->C;
  if (p2->C != C || C > PXO_MAXC) return -1;
  double f1[PXO_MAXC], f2[PXO_MAXC], g1x[PXO_MAXC], g1y[PXO_MAXC], g2x[PXO_MAXC], g2y[PXO_MAXC];
  const int wantJ = J1 || J2;
  pxo_patch_eval(p1, kp1, cfg, f1, wantJ ? g1x : NULL, wantJ ? g1y : NULL);
  pxo_patch_eval(p2, kp2, cfg, f2, wantJ ? g2x : NULL, wantJ ? g2y : NULL);
  for (int i = 0; i < C; ++i) {
    r[i] = f1[i] - f2[i];
    if (J1) { J1[i * 2] = g1x[i]; J1[i * 2 + 1] = g1y[i]; }
    if (J2) { J2[i * 2] = -g2x[i]; J2[i * 2 + 1] = -g2y[i]; }
  }
  return 1;
}

/* FeatureReference2DCostFunctor::operator() (residuals/src/feature_reference.h:44-60). */
int pxo_ref2d_residual(const pxo_patch* p, const pxo_interp_cfg* cfg, const double kp[2],
                       const double* ref, double* r, double* J) {
  const int C = p->C;
  if (C > PXO_MAXC) return -1;
  double gx[PXO_MAXC], gy[PXO_MAXC];
  pxo_patch_eval(p, kp, cfg, r, J ? gx : NULL, J ? gy : NULL);
  for (int i = 0; i < C; ++i) {
    r[i] -= ref[i];
    if (J) { J[i * 2] = gx[i]; J[i * 2 + 1] = gy[i]; }
  }
  return 1;
}

/* [upstream Ceres 2.1 loss_function.cc] TrivialLoss / HuberLoss / SoftLOneLoss /
 * CauchyLoss::Evaluate and ScaledLoss (used at
 * keypoint_adjustment/src/featuremetric_keypoint_optimizer.h:193-194).               */
void pxo_loss_eval(const pxo_loss* loss, double weight, double s, double rho[3]) {
  const double a = loss->a, b = a * a;
  switch (loss->type) {
    case PXO_LOSS_CAUCHY: {
      const double c = 1.0 / b;
      const double sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * log(sum);
      rho[1] = inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308;
      rho[2] = -c * (inv * inv);
      break;
    }
    case PXO_LOSS_HUBER:
      if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = a / r > 2.2250738585072014e-308 ? a / r : 2.2250738585072014e-308;
        rho[2] = -rho[1] / (2.0 * s);
      } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
      }
      break;
    case PXO_LOSS_SOFTL1: {
      const double c = 1.0 / b;
      const double sum = 1.0 + s * c, tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0);
      rho[1] = 1.0 / tmp > 2.2250738585072014e-308 ? 1.0 / tmp : 2.2250738585072014e-308;
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      break;
    }
    default:
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
  rho[0] *= weight; rho[1] *= weight; rho[2] *= weight;
}

/* [upstream Ceres 2.1 corrector.cc] Corrector::Corrector / CorrectResiduals /
 * CorrectJacobian.                                                                    */
void pxo_corrector(double s, const double rho[3], int C, int n, double* r, double* J) {
  const double sqrt_rho1 = sqrt(rho[1]);
  double residual_scaling, alpha_sq_norm;
  if (s == 0.0 || rho[2] <= 0.0) {
    residual_scaling = sqrt_rho1;
    alpha_sq_norm = 0.0;
  } else {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - sqrt(D);
    residual_scaling = sqrt_rho1 / (1 - alpha);
    alpha_sq_norm = alpha / s;
  }
  if (J) {
    if (alpha_sq_norm == 0.0) {
      for (int i = 0; i < C * n; ++i) J[i] *= sqrt_rho1;
    } else {
      for (int c = 0; c < n; ++c) {
        double rtj = 0;
        for (int i = 0; i < C; ++i) rtj += J[i * n + c] * r[i];
        for (int i = 0; i < C; ++i)
          J[i * n + c] = sqrt_rho1 * (J[i * n + c] - alpha_sq_norm * r[i] * rtj);
      }
    }
  }
  for (int i = 0; i < C; ++i) r[i] *= residual_scaling;
}

/* RobustMeanIRLS (base/src/irls_optim.h:24-71), N_NODES = 1, ncc_normalize = false. */
int pxo_robust_mean_irls(const double* descs, int n, int C, const pxo_loss* loss, int iters,
                         int l2_normalize, double* mean) {
  double* w = (double*)malloc(sizeof(double) * n);
  for (int i = 0; i < n; ++i) w[i] = 1.0;
  int early = -1;
  for (int k = 0; k < iters && early < 0; ++k) {
    double sw = 0;
    for (int i = 0; i < n; ++i) sw += w[i];
    for (int i = 0; i < n; ++i) w[i] = w[i] / sw; /* :44 */
    for (int c = 0; c < C; ++c) mean[c] = 0.0;
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < C; ++c) mean[c] += descs[(int64_t)i * C + c] * w[i]; /* :46-48 */
    if (l2_normalize) { /* :54-58 Eigen normalize(): divide by norm */
      double ss = 0;
      for (int c = 0; c < C; ++c) ss += mean[c] * mean[c];
      const double nrm = sqrt(ss);
      if (nrm > 0)
        for (int c = 0; c < C; ++c) mean[c] /= nrm;
    }
    for (int i = 0; i < n; ++i) { /* :60-69 */
      double s = 0;
      for (int c = 0; c < C; ++c) {
        const double d = descs[(int64_t)i * C + c] - mean[c];
        s += d * d;
      }
      double rho[3];
      pxo_loss_eval(loss, 1.0, s, rho);
      if (rho[0] > 0.0) {
        w[i] = 1.0 / rho[0];
      } else {
        early = i;
        memcpy(mean, descs + (int64_t)i * C, sizeof(double) * C);
        break;
      }
    }
  }
  free(w);
  return early;
}

/* ReferenceExtractor::ComputeReference (bundle_adjustment/src/reference_extractor.h:238-272). */
int pxo_compute_reference(const double* descs, int n, int C, const pxo_loss* loss, int iters,
                          int l2_normalize, double* ref_out, double* robust_mean_out) {
  double mean[PXO_MAXC];
  pxo_robust_mean_irls(descs, n, C, loss, iters, l2_normalize, mean);
  int best = 0;
  double bestd = 0;
  for (int i = 0; i < n; ++i) {
    double s = 0;
    for (int c = 0; c < C; ++c) {
      const double d = descs[(int64_t)i * C + c] - mean[c];
      s += d * d;
    }
    if (i == 0 || s < bestd) { bestd = s; best = i; } /* minCoeff: first minimum */
  }
  if (robust_mean_out) memcpy(robust_mean_out, mean, sizeof(double) * C);
  if (ref_out) memcpy(ref_out, descs + (int64_t)best * C, sizeof(double) * C);
  return best;
}

/* ---------------------------------------------------------------------------------
 * Batch evaluator: what Ceres does per residual block in one Jacobian evaluation
 * (materialised 128 x n row-major blocks, loss + corrector), threaded over blocks like
 * bundle_adjustment_options.h:58 (num_threads = -1).  Used for tests and for the
 * cpu_baseline leg of bench.py.
 * --------------------------------------------------------------------------------- */
typedef struct {
  const pxo_ba_batch* b; const pxo_interp_cfg* cfg; const pxo_loss* loss;
  int64_t first, count; double* r_out; double* J_out; double cost;
} ba_job;

static void* ba_worker(void* arg) {
  ba_job* j = (ba_job*)arg;
  const pxo_ba_batch* b = j->b;
  const int C = b->C;
  const int NJ = 10 + PXO_KPAD;
  const size_t es = b->dtype == PXO_F16 ? 2 : (b->dtype == PXO_F32 ? 4 : 8);
  double r[PXO_MAXC];
  double* Jq = (double*)malloc(sizeof(double) * C * (4 + 3 + 3 + PXO_KPAD));
  double* Jt = Jq + C * 4; double* JX = Jt + C * 3; double* Jk = JX + C * 3;
  double cost = 0;
  for (int64_t i = j->first; i < j->first + j->count; ++i) {
    const int img = b->obs_image[i], pt = b->obs_point[i], cam = b->image_camera[img];
    const int64_t pi = b->obs_patch[i];
    pxo_patch p;
    p.data = (const char*)b->arena + (size_t)pi * b->H * b->W * C * es;
    p.dtype = b->dtype; p.H = b->H; p.W = b->W; p.C = C;
    p.x0 = b->corners[2 * pi]; p.y0 = b->corners[2 * pi + 1];
    p.sx = b->scales[2 * pi]; p.sy = b->scales[2 * pi + 1]; p.up = b->upsampling > 0.0 ? b->upsampling : 1.0;
    const int model = b->cam_model[cam];
    const int K = pxo_camera_num_params(model);
    pxo_ba_residual(&p, j->cfg, model, b->qvec + 4 * img, b->tvec + 3 * img, b->xyz + 3 * (int64_t)pt,
                    b->cam_params + PXO_KPAD * cam, b->refs ? b->refs + (int64_t)C * pt : NULL, r, Jq, Jt, JX, Jk);
    double s = 0;
    for (int c = 0; c < C; ++c) s += r[c] * r[c];
    double rho[3];
    pxo_loss_eval(j->loss, 1.0, s, rho);
    cost += 0.5 * rho[0];
    if (j->r_out) memcpy(j->r_out + (i - 0) * C, r, sizeof(double) * C);
    if (j->J_out) {
      double* Jo = j->J_out + (int64_t)i * C * NJ;
      for (int c = 0; c < C; ++c) {
        double* row = Jo + c * NJ;
        memcpy(row, Jq + c * 4, 32); memcpy(row + 4, Jt + c * 3, 24); memcpy(row + 7, JX + c * 3, 24);
        memset(row + 10, 0, sizeof(double) * PXO_KPAD);
        memcpy(row + 10, Jk + c * K, sizeof(double) * K);
      }
    }
  }
  free(Jq);
  j->cost = cost;
  return NULL;
}

double pxo_ba_eval_batch(const pxo_ba_batch* b, const pxo_interp_cfg* cfg, const pxo_loss* loss,
                         int64_t first, int64_t count, int n_threads, double* r_out,
                         double* J_out) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  ba_job jobs[256];
  pthread_t th[256];
  /* r_out / J_out are indexed relative to `first` */
  int64_t per = (count + n_threads - 1) / n_threads;
  int used = 0;
  for (int t = 0; t < n_threads; ++t) {
    int64_t lo = first + t * per, hi = lo + per;
    if (hi > first + count) hi = first + count;
    if (lo >= hi) break;
    jobs[t].b = b; jobs[t].cfg = cfg; jobs[t].loss = loss;
    jobs[t].first = lo; jobs[t].count = hi - lo;
    jobs[t].r_out = r_out ? r_out - first * b->C : NULL;
    jobs[t].J_out = J_out ? J_out - first * (int64_t)b->C * (10 + PXO_KPAD) : NULL;
    jobs[t].cost = 0;
    ++used;
  }
  if (used == 1) {
    ba_worker(&jobs[0]);
  } else {
    for (int t = 0; t < used; ++t) pthread_create(&th[t], NULL, ba_worker, &jobs[t]);
    for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
  }
  double cost = 0;
  for (int t = 0; t < used; ++t) cost += jobs[t].cost;
  return cost;
}

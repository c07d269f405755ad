/*
 * pxo_solve.c -- ORACLE (test infrastructure only; see pxo.h header).
 *
 * Dense-matrix restatement of the solves the reference delegates to Ceres:
 *   BA: BundleOptimizer::SolveProblem (bundle_adjustment/src/bundle_optimizer.h:172-245) with
 *       the parameterisation of :335-453 (quaternion manifold, subset manifolds, constant
 *       blocks), residual blocks from feature_reference_bundle_optimizer.h:90-149;
 *   KA: KeypointOptimizerBase::SolveProblem (keypoint_adjustment/src/keypoint_optimizer.h:77-104)
 *       with the box bounds of :110-157 and the edges of
 *       topological_keypoint_optimizer.h:97-175 / featuremetric_keypoint_optimizer.h:158-202.
 *
 * [upstream Ceres 2.1] The trust-region Levenberg-Marquardt loop is restated from
 * trust_region_minimizer.cc / levenberg_marquardt_strategy.cc: Jacobi column scaling
 * 1/(1+||J_j||) fixed at iteration 0; LM diagonal sqrt(clamp(diag(J^T J), 1e-6, 1e32)/radius);
 * model_cost_change = -(J d).(r + J d / 2); accept if rho > 1e-3; radius <- radius /
 * max(1/3, 1 - (2 rho - 1)^3) on success, radius /= decrease_factor (2, 4, 8 ...) on failure;
 * parameter tolerance |dx| <= tol (|x| + tol); function tolerance |dcost| <= tol * cost.
 * Unlike the reference's SPARSE_SCHUR / SPARSE_NORMAL_CHOLESKY the oracle factorises the full
 * damped normal matrix densely -- mathematically the same step, independent code path.
 * PARITY UNPINNED: Ceres is not available, so trajectories cannot be compared with the real
 * reference; the GPU solver is compared with THIS restatement.  Inner iterations
 * (bundle_adjustment/main.py:43) and the bounds line search are restated in simplified
 * form (see the functions below).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pxo.h"

#define PXO_MAXC 512
#define PXO_KPAD 12

/* ------------------------------------------------------------------------------------ */
/* dense symmetric positive definite solve (Cholesky), returns 0 on success              */
static int chol_solve(int n, double* A /* n x n row-major, overwritten (lower) */, double* b) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0) || !isfinite(d)) return -1;
    d = sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  return 0;
}

/* [upstream Ceres 2.1 manifold.cc] QuaternionManifold::Plus: x+ = [cos|d|, sin|d|/|d| d] * x */
static void quat_plus(const double* x, const double* d, double* xp) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { memcpy(xp, x, 32); return; }
  const double s = sin(nd) / nd;
  const double q[4] = {cos(nd), s * d[0], s * d[1], s * d[2]};
  xp[0] = q[0] * x[0] - q[1] * x[1] - q[2] * x[2] - q[3] * x[3];
  xp[1] = q[0] * x[1] + q[1] * x[0] + q[2] * x[3] - q[3] * x[2];
  xp[2] = q[0] * x[2] - q[1] * x[3] + q[2] * x[0] + q[3] * x[1];
  xp[3] = q[0] * x[3] + q[1] * x[2] - q[2] * x[1] + q[3] * x[0];
}
/* QuaternionManifold::PlusJacobian (4 x 3 row-major) */
static void quat_plus_jac(const double* x, double* J) {
  J[0] = -x[1]; J[1] = -x[2]; J[2] = -x[3];
  J[3] = x[0];  J[4] = x[3];  J[5] = -x[2];
  J[6] = -x[3]; J[7] = x[0];  J[8] = x[1];
  J[9] = x[2];  J[10] = -x[1]; J[11] = x[0];
}

/* ------------------------------------------------------------------------------------ */
typedef struct {
  /* unknown layout */
  int n;                 /* total tangent unknowns */
  int* pose_off; int* pose_dim; /* per image: offset of [3 rot][free tvec comps] */
  int* intr_off; int* intr_dim; /* per camera */
  int* pt_off;                  /* per point: offset or -1 */
} ba_layout;

typedef struct {
  pxo_ba_batch* b; const pxo_interp_cfg* cfg; const pxo_loss* loss;
  const uint8_t* pose_const; const uint8_t* tvec_const_mask; const uint16_t* cam_const_mask;
  const uint8_t* point_const;
  int n_images, n_cams; int64_t n_points;
  ba_layout L;
} ba_ctx;

static void ba_build_layout(ba_ctx* c) {
  ba_layout* L = &c->L;
  L->pose_off = (int*)malloc(sizeof(int) * c->n_images);
  L->pose_dim = (int*)malloc(sizeof(int) * c->n_images);
  L->intr_off = (int*)malloc(sizeof(int) * c->n_cams);
  L->intr_dim = (int*)malloc(sizeof(int) * c->n_cams);
  L->pt_off = (int*)malloc(sizeof(int) * c->n_points);
  /* only blocks that appear in at least one residual are part of the program */
  uint8_t* img_used = (uint8_t*)calloc(c->n_images, 1);
  uint8_t* cam_used = (uint8_t*)calloc(c->n_cams, 1);
  uint8_t* pt_used = (uint8_t*)calloc(c->n_points, 1);
  for (int64_t i = 0; i < c->b->n_obs; ++i) {
    img_used[c->b->obs_image[i]] = 1;
    cam_used[c->b->image_camera[c->b->obs_image[i]]] = 1;
    pt_used[c->b->obs_point[i]] = 1;
  }
  int off = 0;
  for (int i = 0; i < c->n_images; ++i) {
    int d = 0;
    if (img_used[i] && !c->pose_const[i]) {
      d = 3;
      for (int k = 0; k < 3; ++k) if (!((c->tvec_const_mask[i] >> k) & 1)) ++d;
    }
    L->pose_off[i] = off; L->pose_dim[i] = d; off += d;
  }
  for (int j = 0; j < c->n_cams; ++j) {
    int d = 0;
    if (cam_used[j]) {
      const int K = pxo_camera_num_params(c->b->cam_model[j]);
      for (int k = 0; k < K; ++k) if (!((c->cam_const_mask[j] >> k) & 1)) ++d;
    }
    L->intr_off[j] = off; L->intr_dim[j] = d; off += d;
  }
  for (int64_t p = 0; p < c->n_points; ++p) {
    if (pt_used[p] && !c->point_const[p]) { L->pt_off[p] = off; off += 3; } else L->pt_off[p] = -1;
  }
  L->n = off;
  free(img_used); free(cam_used); free(pt_used);
}

static pxo_patch ba_patch(const pxo_ba_batch* b, int64_t pi) {
  const size_t es = b->dtype == PXO_F16 ? 2 : (b->dtype == PXO_F32 ? 4 : 8);
  pxo_patch p;
  p.data = (const char*)b->arena + (size_t)pi * b->H * b->W * b->C * es;
  p.dtype = b->dtype; p.H = b->H; p.W = b->W; p.C = b->C;
  p.x0 = b->corners[2 * pi]; p.y0 = b->corners[2 * pi + 1];
  p.sx = b->scales[2 * pi]; p.sy = b->scales[2 * pi + 1]; p.up = b->upsampling > 0.0 ? b->upsampling : 1.0;
  return p;
}

/* cost (and optionally H = J~^T J~ (n x n, full), g = J~^T r~ in UNSCALED tangent space) */
static double ba_evaluate(ba_ctx* c, const double* qvec, const double* tvec, const double* cams,
                          const double* xyz, double* H, double* g) {
  const pxo_ba_batch* b = c->b;
  const int C = b->C, n = c->L.n;
  double cost = 0;
  if (H) { memset(H, 0, sizeof(double) * (size_t)n * n); memset(g, 0, sizeof(double) * n); }
  double* r = (double*)malloc(sizeof(double) * C);
  double* Jq = (double*)malloc(sizeof(double) * C * (4 + 3 + 3 + PXO_KPAD));
  double *Jt = Jq + C * 4, *JX = Jt + C * 3, *Jk = JX + C * 3;
  double* Jl = (double*)malloc(sizeof(double) * C * 24); /* local tangent Jacobian C x dl */
  int idx[24];
  for (int64_t i = 0; i < b->n_obs; ++i) {
    const int img = b->obs_image[i], cam = b->image_camera[img];
    const int64_t pt = b->obs_point[i];
    const int model = b->cam_model[cam], K = pxo_camera_num_params(model);
    pxo_patch p = ba_patch(b, b->obs_patch[i]);
    pxo_ba_residual(&p, c->cfg, model, qvec + 4 * img, tvec + 3 * img, xyz + 3 * pt,
                    cams + PXO_KPAD * cam, b->refs ? b->refs + (size_t)C * pt : NULL, r, H ? Jq : NULL, H ? Jt : NULL,
                    H ? JX : NULL, H ? Jk : NULL);
    double s = 0;
    for (int k = 0; k < C; ++k) s += r[k] * r[k];
    double rho[3];
    pxo_loss_eval(c->loss, 1.0, s, rho);
    cost += 0.5 * rho[0];
    if (!H) continue;
    /* ambient -> tangent columns + global indices */
    int dl = 0;
    if (c->L.pose_dim[img] > 0) {
      double PJ[12];
      quat_plus_jac(qvec + 4 * img, PJ);
      for (int a = 0; a < 3; ++a) {
        for (int k = 0; k < C; ++k) {
          double v = 0;
          for (int m = 0; m < 4; ++m) v += Jq[k * 4 + m] * PJ[m * 3 + a];
          Jl[k * 24 + dl] = v;
        }
        idx[dl] = c->L.pose_off[img] + a; ++dl;
      }
      int tc = 0;
      for (int a = 0; a < 3; ++a) {
        if ((c->tvec_const_mask[img] >> a) & 1) continue;
        for (int k = 0; k < C; ++k) Jl[k * 24 + dl] = Jt[k * 3 + a];
        idx[dl] = c->L.pose_off[img] + 3 + tc; ++tc; ++dl;
      }
    }
    if (c->L.intr_dim[cam] > 0) {
      int kc = 0;
      for (int a = 0; a < K; ++a) {
        if ((c->cam_const_mask[cam] >> a) & 1) continue;
        for (int k = 0; k < C; ++k) Jl[k * 24 + dl] = Jk[k * K + a];
        idx[dl] = c->L.intr_off[cam] + kc; ++kc; ++dl;
      }
    }
    if (c->L.pt_off[pt] >= 0) {
      for (int a = 0; a < 3; ++a) {
        for (int k = 0; k < C; ++k) Jl[k * 24 + dl] = JX[k * 3 + a];
        idx[dl] = c->L.pt_off[pt] + a; ++dl;
      }
    }
    /* corrector on (r, Jl) -- Jl has row stride 24; compact to C x dl first */
    double* Jc = (double*)malloc(sizeof(double) * C * (dl > 0 ? dl : 1));
    for (int k = 0; k < C; ++k) for (int a = 0; a < dl; ++a) Jc[k * dl + a] = Jl[k * 24 + a];
    pxo_corrector(s, rho, C, dl, r, Jc);
    for (int a = 0; a < dl; ++a) {
      double ga = 0;
      for (int k = 0; k < C; ++k) ga += Jc[k * dl + a] * r[k];
      g[idx[a]] += ga;
      for (int bb = 0; bb < dl; ++bb) {
        double h = 0;
        for (int k = 0; k < C; ++k) h += Jc[k * dl + a] * Jc[k * dl + bb];
        H[(size_t)idx[a] * n + idx[bb]] += h;
      }
    }
    free(Jc);
  }
  free(r); free(Jq); free(Jl);
  return cost;
}

/* x (+) delta for all blocks (delta in unscaled tangent space) */
static void ba_plus(ba_ctx* c, const double* q0, const double* t0, const double* k0, const double* X0,
                    const double* delta, double* q1, double* t1, double* k1, double* X1) {
  memcpy(q1, q0, sizeof(double) * 4 * c->n_images);
  memcpy(t1, t0, sizeof(double) * 3 * c->n_images);
  memcpy(k1, k0, sizeof(double) * PXO_KPAD * c->n_cams);
  memcpy(X1, X0, sizeof(double) * 3 * c->n_points);
  for (int i = 0; i < c->n_images; ++i) {
    if (c->L.pose_dim[i] == 0) continue;
    const double* d = delta + c->L.pose_off[i];
    quat_plus(q0 + 4 * i, d, q1 + 4 * i);
    int tc = 0;
    for (int a = 0; a < 3; ++a) {
      if ((c->tvec_const_mask[i] >> a) & 1) continue;
      t1[3 * i + a] = t0[3 * i + a] + d[3 + tc]; ++tc;
    }
  }
  for (int j = 0; j < c->n_cams; ++j) {
    if (c->L.intr_dim[j] == 0) continue;
    const int K = pxo_camera_num_params(c->b->cam_model[j]);
    int kc = 0;
    for (int a = 0; a < K; ++a) {
      if ((c->cam_const_mask[j] >> a) & 1) continue;
      k1[PXO_KPAD * j + a] = k0[PXO_KPAD * j + a] + delta[c->L.intr_off[j] + kc]; ++kc;
    }
  }
  for (int64_t p = 0; p < c->n_points; ++p) {
    if (c->L.pt_off[p] < 0) continue;
    for (int a = 0; a < 3; ++a) X1[3 * p + a] = X0[3 * p + a] + delta[c->L.pt_off[p] + a];
  }
}

/* squared ambient norms over the variable blocks (Ceres' x_norm / step_norm) */
static double ba_ambient_sqnorm(ba_ctx* c, const double* q, const double* t, const double* k,
                                const double* X, const double* q2, const double* t2,
                                const double* k2, const double* X2) {
  double s = 0;
#define DIFF(a, b, i) ((a)[i] - ((b) ? (b)[i] : 0.0))
  for (int i = 0; i < c->n_images; ++i) {
    if (c->L.pose_dim[i] == 0) continue;
    for (int a = 0; a < 4; ++a) { double d = DIFF(q, q2, 4 * i + a); s += d * d; }
    /* tvec: SubsetManifold keeps the whole block in the ambient vector */
    for (int a = 0; a < 3; ++a) { double d = DIFF(t, t2, 3 * i + a); s += d * d; }
  }
  for (int j = 0; j < c->n_cams; ++j) {
    if (c->L.intr_dim[j] == 0) continue;
    const int K = pxo_camera_num_params(c->b->cam_model[j]);
    for (int a = 0; a < K; ++a) { double d = DIFF(k, k2, PXO_KPAD * j + a); s += d * d; }
  }
  for (int64_t p = 0; p < c->n_points; ++p) {
    if (c->L.pt_off[p] < 0) continue;
    for (int a = 0; a < 3; ++a) { double d = DIFF(X, X2, 3 * p + a); s += d * d; }
  }
#undef DIFF
  return s;
}

/* ---- inner iterations ---------------------------------------------------------------------
 * [upstream Ceres 2.1 coordinate_descent_minimizer.cc + trust_region_minimizer.cc
 * DoInnerIterationsIfNeeded]  pixsfm enables them by default for BA
 * (bundle_adjustment/main.py:43) with all variable points in group 0
 * (bundle_optimizer.h:350-355): after the trust-region step every variable point is re-optimised
 * on its own (cameras fixed at the candidate) by a nested TR-LM with Ceres' DEFAULT solver
 * options -- 50 iterations, function/gradient/parameter tolerance 1e-6 / 1e-10 / 1e-8, initial
 * radius 1e4, Jacobi scaling, at most 5 consecutive invalid steps.
 * cost_p(X) = sum over the point's observations of 0.5 rho(|r|^2). */
static double point_eval(ba_ctx* c, const double* q, const double* t, const double* k, int64_t p,
                         const int64_t* obs, int n_obs, const double* X, double* H, double* g) {
  const pxo_ba_batch* b = c->b;
  const int C = b->C;
  double cost = 0, r[PXO_MAXC], JX[PXO_MAXC * 3];
  if (H) { memset(H, 0, sizeof(double) * 9); memset(g, 0, sizeof(double) * 3); }
  for (int o = 0; o < n_obs; ++o) {
    const int64_t i = obs[o];
    const int img = b->obs_image[i], cam = b->image_camera[img];
    pxo_patch pt = ba_patch(b, b->obs_patch[i]);
    pxo_ba_residual(&pt, c->cfg, b->cam_model[cam], q + 4 * img, t + 3 * img, X, k + PXO_KPAD * cam,
                    b->refs ? b->refs + (size_t)C * p : NULL, r, NULL, NULL, H ? JX : NULL, NULL);
    double s = 0;
    for (int m = 0; m < C; ++m) s += r[m] * r[m];
    double rho[3];
    pxo_loss_eval(c->loss, 1.0, s, rho);
    cost += 0.5 * rho[0];
    if (!H) continue;
    pxo_corrector(s, rho, C, 3, r, JX);
    for (int a = 0; a < 3; ++a) {
      for (int m = 0; m < C; ++m) g[a] += JX[m * 3 + a] * r[m];
      for (int bb = 0; bb < 3; ++bb) {
        double h = 0;
        for (int m = 0; m < C; ++m) h += JX[m * 3 + a] * JX[m * 3 + bb];
        H[a * 3 + bb] += h;
      }
    }
  }
  return cost;
}

static void point_inner_lm(ba_ctx* c, const double* q, const double* t, const double* k, int64_t p,
                           const int64_t* obs, int n_obs, double* X) {
  double H[9], A[9], g[3], scale[3], diag[3], step[3], Xc[3];
  double cost = point_eval(c, q, t, k, p, obs, n_obs, X, H, g);
  double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  if (gmax <= 1e-10) return;
  for (int j = 0; j < 3; ++j) scale[j] = 1.0 / (1.0 + sqrt(H[j * 3 + j]));
#define PT_SCALE() do { for (int a = 0; a < 3; ++a) { g[a] *= scale[a]; for (int bb = 0; bb < 3; ++bb) H[a * 3 + bb] *= scale[a] * scale[bb]; } } while (0)
  PT_SCALE();
  double radius = 1e4, decrease_factor = 2.0;
  int invalid = 0, reuse_diag = 0;
  for (int it = 0; it < 50; ++it) {
    if (radius < 1e-32) break;
    if (!reuse_diag)
      for (int j = 0; j < 3; ++j) diag[j] = fmin(fmax(H[j * 3 + j], 1e-6), 1e32);
    memcpy(A, H, sizeof(A));
    for (int j = 0; j < 3; ++j) { A[j * 3 + j] += diag[j] / radius; step[j] = -g[j]; }
    int ok = chol_solve(3, A, step) == 0;
    double mcc = 0;
    if (ok) {
      double dg = 0, dHd = 0;
      for (int a = 0; a < 3; ++a) {
        dg += step[a] * g[a];
        for (int bb = 0; bb < 3; ++bb) dHd += step[a] * H[a * 3 + bb] * step[bb];
        if (!isfinite(step[a])) ok = 0;
      }
      mcc = -dg - 0.5 * dHd;
      if (!(mcc > 0.0)) ok = 0;
    }
    if (!ok) {
      if (++invalid >= 5) break;
      radius *= 0.5; reuse_diag = 1;
      continue;
    }
    invalid = 0;
    for (int j = 0; j < 3; ++j) Xc[j] = X[j] + step[j] * scale[j];
    const double cand = point_eval(c, q, t, k, p, obs, n_obs, Xc, NULL, NULL);
    double s2 = 0, x2 = 0;
    for (int j = 0; j < 3; ++j) { s2 += (Xc[j] - X[j]) * (Xc[j] - X[j]); x2 += X[j] * X[j]; }
    if (sqrt(s2) <= 1e-8 * (sqrt(x2) + 1e-8)) break;
    const double cost_change = cost - cand;
    if (fabs(cost_change) <= 1e-6 * cost) break;
    const double rel = cost_change / mcc;
    if (rel > 1e-3) {
      memcpy(X, Xc, sizeof(Xc));
      cost = point_eval(c, q, t, k, p, obs, n_obs, X, H, g);
      gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
      PT_SCALE();
      const double tmp = 2.0 * rel - 1.0;
      radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp));
      decrease_factor = 2.0; reuse_diag = 0;
      if (gmax <= 1e-10) break;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diag = 1;
    }
  }
#undef PT_SCALE
}

/* re-optimise every variable point in place (X1), cameras fixed */
static void ba_inner_iterations(ba_ctx* c, const double* q1, const double* t1, const double* k1, double* X1) {
  const pxo_ba_batch* b = c->b;
  int64_t* cnt = (int64_t*)calloc(c->n_points + 1, sizeof(int64_t));
  for (int64_t i = 0; i < b->n_obs; ++i) ++cnt[b->obs_point[i] + 1];
  for (int64_t p = 0; p < c->n_points; ++p) cnt[p + 1] += cnt[p];
  int64_t* lst = (int64_t*)malloc(sizeof(int64_t) * (b->n_obs ? b->n_obs : 1));
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (c->n_points + 1));
  memcpy(cur, cnt, sizeof(int64_t) * (c->n_points + 1));
  for (int64_t i = 0; i < b->n_obs; ++i) lst[cur[b->obs_point[i]]++] = i;
  for (int64_t p = 0; p < c->n_points; ++p) {
    if (c->L.pt_off[p] < 0 || cnt[p + 1] == cnt[p]) continue;
    point_inner_lm(c, q1, t1, k1, p, lst + cnt[p], (int)(cnt[p + 1] - cnt[p]), X1 + 3 * p);
  }
  free(cnt); free(lst); free(cur);
}

int pxo_ba_solve(pxo_ba_batch* b, int n_images, int n_cams, int64_t n_points,
                 const pxo_interp_cfg* cfg, const pxo_loss* loss, const uint8_t* pose_const,
                 const uint8_t* tvec_const_mask, const uint16_t* cam_const_mask,
                 const uint8_t* point_const, const pxo_lm_options* opt, pxo_lm_summary* sum) {
  ba_ctx c;
  memset(&c, 0, sizeof(c));
  c.b = b; c.cfg = cfg; c.loss = loss; c.pose_const = pose_const; c.tvec_const_mask = tvec_const_mask;
  c.cam_const_mask = cam_const_mask; c.point_const = point_const;
  c.n_images = n_images; c.n_cams = n_cams; c.n_points = n_points;
  ba_build_layout(&c);
  const int n = c.L.n;
  double* q = (double*)b->qvec; double* t = (double*)b->tvec;
  double* k = (double*)b->cam_params; double* X = (double*)b->xyz;
  /* AddImageToProblem normalises the quaternions (bundle_optimizer.h:255) */
  for (int i = 0; i < n_images; ++i) {
    double nn = sqrt(q[4 * i] * q[4 * i] + q[4 * i + 1] * q[4 * i + 1] + q[4 * i + 2] * q[4 * i + 2] + q[4 * i + 3] * q[4 * i + 3]);
    for (int a = 0; a < 4; ++a) q[4 * i + a] /= nn;
  }
  double* H = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* g = (double*)malloc(sizeof(double) * n);
  double* scale = (double*)malloc(sizeof(double) * n);
  double* step = (double*)malloc(sizeof(double) * n);
  double* delta = (double*)malloc(sizeof(double) * n);
  double* diag = (double*)malloc(sizeof(double) * n);
  double* q1 = (double*)malloc(sizeof(double) * 4 * n_images);
  double* t1 = (double*)malloc(sizeof(double) * 3 * n_images);
  double* k1 = (double*)malloc(sizeof(double) * PXO_KPAD * n_cams);
  double* X1 = (double*)malloc(sizeof(double) * 3 * n_points);

  double cost = ba_evaluate(&c, q, t, k, X, H, g);
  sum->initial_cost = cost; sum->num_unknowns = n;
  sum->iterations = 0; sum->num_successful = 0; sum->termination = PXO_TERM_NO_CONVERGENCE;
  for (int j = 0; j < n; ++j) scale[j] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[(size_t)j * n + j])) : 1.0;
  /* scale the system: Hs = S H S, gs = S g */
#define SCALE_SYSTEM()                                                                   \
  do {                                                                                   \
    for (int a = 0; a < n; ++a) {                                                        \
      g[a] *= scale[a];                                                                  \
      for (int bb = 0; bb < n; ++bb) H[(size_t)a * n + bb] *= scale[a] * scale[bb];     \
    }                                                                                    \
  } while (0)
  SCALE_SYSTEM();
  double radius = opt->initial_radius, decrease_factor = 2.0;
  int invalid = 0;
  int reuse_diag = 0;
  int inner_enabled = opt->use_inner_iterations;
  while (1) {
    if (sum->iterations >= opt->max_iterations) { sum->termination = PXO_TERM_NO_CONVERGENCE; break; }
    if (radius < opt->min_radius) { sum->termination = PXO_TERM_CONVERGENCE; break; }
    ++sum->iterations;
    /* LevenbergMarquardtStrategy::ComputeStep */
    if (!reuse_diag)
      for (int j = 0; j < n; ++j) {
        double d = H[(size_t)j * n + j];
        diag[j] = d < opt->min_lm_diagonal ? opt->min_lm_diagonal : (d > opt->max_lm_diagonal ? opt->max_lm_diagonal : d);
      }
    memcpy(A, H, sizeof(double) * (size_t)n * n);
    for (int j = 0; j < n; ++j) { A[(size_t)j * n + j] += diag[j] / radius; step[j] = -g[j]; }
    int ok = chol_solve(n, A, step) == 0;
    double model_cost_change = 0;
    if (ok) {
      /* -(J d).(r + J d/2) = -d.g - 0.5 d.H.d */
      double dg = 0, dHd = 0;
      for (int a = 0; a < n; ++a) {
        dg += step[a] * g[a];
        double hr = 0;
        for (int bb = 0; bb < n; ++bb) hr += H[(size_t)a * n + bb] * step[bb];
        dHd += step[a] * hr;
        if (!isfinite(step[a])) ok = 0;
      }
      model_cost_change = -dg - 0.5 * dHd;
      if (!(model_cost_change > 0.0)) ok = 0;
    }
    if (!ok) { /* HandleInvalidStep */
      if (++invalid >= opt->max_consecutive_invalid_steps) { sum->termination = PXO_TERM_FAILURE; break; }
      radius *= 0.5; reuse_diag = 1;   /* StepIsInvalid */
      continue;
    }
    invalid = 0;
    for (int j = 0; j < n; ++j) delta[j] = step[j] * scale[j];
    ba_plus(&c, q, t, k, X, delta, q1, t1, k1, X1);
    double cand = ba_evaluate(&c, q1, t1, k1, X1, NULL, NULL);
    int inner_useful = 0;
    if (inner_enabled && isfinite(cand)) { /* DoInnerIterationsIfNeeded [upstream trust_region_minimizer.cc] */
      ba_inner_iterations(&c, q1, t1, k1, X1);
      const double inner_cost = ba_evaluate(&c, q1, t1, k1, X1, NULL, NULL);
      model_cost_change += cand - inner_cost;
      inner_useful = inner_cost < cost;
      inner_enabled = (1.0 - inner_cost / cand) > opt->inner_iteration_tolerance;
      cand = inner_cost;
    }
    const double step_norm = sqrt(ba_ambient_sqnorm(&c, q, t, k, X, q1, t1, k1, X1));
    const double x_norm = sqrt(ba_ambient_sqnorm(&c, q, t, k, X, NULL, NULL, NULL, NULL));
    if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
      sum->termination = PXO_TERM_CONVERGENCE; break;
    }
    const double cost_change = cost - cand;
    if (fabs(cost_change) <= opt->function_tolerance * cost) { sum->termination = PXO_TERM_CONVERGENCE; break; }
    const double rel = cost_change / model_cost_change;
    if (inner_useful || rel > opt->min_relative_decrease) { /* IsStepSuccessful + HandleSuccessfulStep */
      memcpy(q, q1, sizeof(double) * 4 * n_images); memcpy(t, t1, sizeof(double) * 3 * n_images);
      memcpy(k, k1, sizeof(double) * PXO_KPAD * n_cams); memcpy(X, X1, sizeof(double) * 3 * n_points);
      cost = ba_evaluate(&c, q, t, k, X, H, g);
      SCALE_SYSTEM();
      ++sum->num_successful;
      double gmax = 0;
      for (int j = 0; j < n; ++j) { double v = fabs(g[j] / scale[j]); if (v > gmax) gmax = v; }
      const double tmp = 2.0 * rel - 1.0;
      double f = 1.0 - tmp * tmp * tmp;
      if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      radius = radius / f;
      if (radius > opt->max_radius) radius = opt->max_radius;
      decrease_factor = 2.0; reuse_diag = 0;
      if (gmax <= opt->gradient_tolerance) { sum->termination = PXO_TERM_CONVERGENCE; break; }
    } else { /* StepRejected */
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = 1;
    }
  }
  sum->final_cost = cost; sum->final_radius = radius;
  free(H); free(A); free(g); free(scale); free(step); free(delta); free(diag);
  free(q1); free(t1); free(k1); free(X1);
  free(c.L.pose_off); free(c.L.pose_dim); free(c.L.intr_off); free(c.L.intr_dim); free(c.L.pt_off);
  return 0;
}

/* ======================================================================================
 * Keypoint adjustment (KA)
 * ====================================================================================== */

/* ParameterizeKeypoints box (keypoint_adjustment/src/keypoint_optimizer.h:127-152). */
static void ka_bounds(const pxo_ka_batch* b, int64_t node, double bound, double lo[2], double hi[2]) {
  const int64_t pi = b->node_patch[node];
  const double sx = b->scales[2 * pi], sy = b->scales[2 * pi + 1];
  const double* kp = b->kp + 2 * node;
  lo[0] = (b->corners[2 * pi] + 0.5) / sx;     lo[1] = (b->corners[2 * pi + 1] + 0.5) / sy;
  hi[0] = lo[0] + b->W / sx;                   hi[1] = lo[1] + b->H / sy;
  if (bound > 0.0) {
    hi[0] = fmin(kp[0] + bound / sx, hi[0]); hi[1] = fmin(kp[1] + bound / sy, hi[1]);
    lo[0] = fmax(kp[0] - bound / sx, lo[0]); lo[1] = fmax(kp[1] - bound / sy, lo[1]);
  }
}

/* the box ParameterizeKeypoints puts around a variable keypoint (keypoint_optimizer.h:124-152), for the parity tests */
void pxo_ka_node_bounds(const pxo_ka_batch* b, int64_t node, double bound, double lo[2], double hi[2]) {
  ka_bounds(b, node, bound, lo, hi);
}

typedef struct {
  const pxo_ka_batch* b; const pxo_interp_cfg* cfg; const pxo_loss* loss;
  const int32_t* edges; int m;    /* edge ids of this problem */
  const int32_t* unary; int nu;   /* unary reference terms of this problem */
  const int* var_of_node;         /* global node -> local unknown offset or -1 */
  int n;                          /* unknowns */
} ka_ctx;

static pxo_patch ka_patch(const pxo_ka_batch* b, int64_t pi) {
  const size_t es = b->dtype == PXO_F16 ? 2 : (b->dtype == PXO_F32 ? 4 : 8);
  pxo_patch p;
  p.data = (const char*)b->arena + (size_t)pi * b->H * b->W * b->C * es;
  p.dtype = b->dtype; p.H = b->H; p.W = b->W; p.C = b->C;
  p.x0 = b->corners[2 * pi]; p.y0 = b->corners[2 * pi + 1];
  p.sx = b->scales[2 * pi]; p.sy = b->scales[2 * pi + 1]; p.up = 1.0;
  return p;
}

/* cost = sum 0.5 w rho(|r|^2); optional dense H = J~^T J~, g = J~^T r~ over the unknowns.
 * kp: full keypoint array (the candidate being evaluated). */
static double ka_evaluate(const ka_ctx* c, const double* kp, double* H, double* g) {
  const pxo_ka_batch* b = c->b;
  const int C = b->C, n = c->n;
  double cost = 0;
  if (H) { memset(H, 0, sizeof(double) * (size_t)n * n); memset(g, 0, sizeof(double) * n); }
  double r[PXO_MAXC], J[PXO_MAXC * 4], J1[PXO_MAXC * 2], J2[PXO_MAXC * 2];
  for (int e = 0; e < c->m; ++e) {
    const int32_t ed = c->edges[e];
    const int64_t n1 = b->edge_src[ed], n2 = b->edge_dst[ed];
    pxo_patch p1 = ka_patch(b, b->node_patch[n1]), p2 = ka_patch(b, b->node_patch[n2]);
    pxo_ka_residual(&p1, &p2, c->cfg, kp + 2 * n1, kp + 2 * n2, r, H ? J1 : NULL, H ? J2 : NULL);
    double s = 0;
    for (int k = 0; k < C; ++k) s += r[k] * r[k];
    double rho[3];
    pxo_loss_eval(c->loss, b->edge_w[ed], s, rho);   /* ScaledLoss, featuremetric_keypoint_optimizer.h:193 */
    cost += 0.5 * rho[0];
    if (!H) continue;
    int idx[4], dl = 0;
    const int v1 = c->var_of_node[n1], v2 = c->var_of_node[n2];
    if (v1 >= 0) { for (int k = 0; k < C; ++k) { J[k * 4 + dl] = J1[k * 2]; J[k * 4 + dl + 1] = J1[k * 2 + 1]; } idx[dl] = v1; idx[dl + 1] = v1 + 1; dl += 2; }
    if (v2 >= 0) { for (int k = 0; k < C; ++k) { J[k * 4 + dl] = J2[k * 2]; J[k * 4 + dl + 1] = J2[k * 2 + 1]; } idx[dl] = v2; idx[dl + 1] = v2 + 1; dl += 2; }
    if (dl == 0) continue;
    double Jc[PXO_MAXC * 4];
    for (int k = 0; k < C; ++k) for (int a = 0; a < dl; ++a) Jc[k * dl + a] = J[k * 4 + a];
    pxo_corrector(s, rho, C, dl, r, Jc);
    for (int a = 0; a < dl; ++a) {
      double ga = 0;
      for (int k = 0; k < C; ++k) ga += Jc[k * dl + a] * r[k];
      g[idx[a]] += ga;
      for (int bb = 0; bb < dl; ++bb) {
        double h = 0;
        for (int k = 0; k < C; ++k) h += Jc[k * dl + a] * Jc[k * dl + bb];
        H[(size_t)idx[a] * n + idx[bb]] += h;
      }
    }
  }
  /* FeatureReference2DCostFunctor blocks (query_keypoint_optimizer.h:122-139) */
  for (int e = 0; e < c->nu; ++e) {
    const int32_t ud = c->unary[e];
    const int64_t nd = b->unary_node[ud];
    pxo_patch p1 = ka_patch(b, b->node_patch[nd]);
    pxo_ref2d_residual(&p1, c->cfg, kp + 2 * nd, b->unary_ref + (size_t)ud * C, r, H ? J1 : NULL);
    double s = 0;
    for (int k = 0; k < C; ++k) s += r[k] * r[k];
    double rho[3];
    pxo_loss_eval(c->loss, b->unary_w ? b->unary_w[ud] : 1.0, s, rho);
    cost += 0.5 * rho[0];
    const int v1 = c->var_of_node[nd];
    if (!H || v1 < 0) continue;
    pxo_corrector(s, rho, C, 2, r, J1);
    for (int a = 0; a < 2; ++a) {
      double ga = 0;
      for (int k = 0; k < C; ++k) ga += J1[k * 2 + a] * r[k];
      g[v1 + a] += ga;
      for (int bb = 0; bb < 2; ++bb) {
        double h = 0;
        for (int k = 0; k < C; ++k) h += J1[k * 2 + a] * J1[k * 2 + bb];
        H[(size_t)(v1 + a) * n + v1 + bb] += h;
      }
    }
  }
  return cost;
}

/* [upstream Ceres 2.1 polynomial.cc] minimise the interpolating polynomial on [lo, hi]:
 * quadratic through (0, f0, g0), (x1, f1); cubic through (0, f0, g0), (x1, f1), (x2, f2). */
static double poly_eval(const double* c, int deg, double x) {
  double v = 0;
  for (int i = deg; i >= 0; --i) v = v * x + c[i];
  return v;
}
static double minimize_poly(const double* c, int deg, double lo, double hi) {
  double best_x = lo, best = poly_eval(c, deg, lo);
  double v = poly_eval(c, deg, hi);
  if (v < best) { best = v; best_x = hi; }
  /* roots of the derivative */
  if (deg == 2) {
    if (c[2] != 0.0) { double x = -c[1] / (2 * c[2]); if (x > lo && x < hi && poly_eval(c, 2, x) < best) { best_x = x; } }
  } else if (deg == 3) {
    const double a = 3 * c[3], bq = 2 * c[2], cq = c[1];
    if (a == 0.0) {
      if (bq != 0.0) { double x = -cq / bq; if (x > lo && x < hi && poly_eval(c, 3, x) < best) best_x = x; }
    } else {
      const double disc = bq * bq - 4 * a * cq;
      if (disc >= 0) {
        const double sq = sqrt(disc);
        const double x1 = (-bq + sq) / (2 * a), x2 = (-bq - sq) / (2 * a);
        if (x1 > lo && x1 < hi) { v = poly_eval(c, 3, x1); if (v < best) { best = v; best_x = x1; } }
        if (x2 > lo && x2 < hi) { v = poly_eval(c, 3, x2); if (v < best) { best = v; best_x = x2; } }
      }
    }
  }
  return best_x;
}
static double interpolating_step(double f0, double g0, int have_prev, double xp, double fp, double xc, double fc,
                                 double lo, double hi) {
  double c[4];
  if (!have_prev) { /* quadratic: c0 = f0, c1 = g0, c2 from f(xc) */
    c[0] = f0; c[1] = g0; c[2] = (fc - f0 - g0 * xc) / (xc * xc);
    return minimize_poly(c, 2, lo, hi);
  }
  /* cubic: f0 + g0 x + a x^2 + b x^3 through (xp, fp), (xc, fc) */
  const double rp = (fp - f0 - g0 * xp) / (xp * xp), rc = (fc - f0 - g0 * xc) / (xc * xc);
  const double bcoef = (rc - rp) / (xc - xp), acoef = rc - bcoef * xc;
  c[0] = f0; c[1] = g0; c[2] = acoef; c[3] = bcoef;
  return minimize_poly(c, 3, lo, hi);
}

static void ka_plus(const ka_ctx* c, const int32_t* nodes, int nn, const double* x0, const double* delta,
                    double alpha, const double* lo, const double* hi, double* x1) {
  /* ParameterBlock::Plus [upstream Ceres]: x + delta then projection onto the box */
  const pxo_ka_batch* b = c->b;
  memcpy(x1, x0, sizeof(double) * 2 * b->n_nodes);
  for (int i = 0; i < nn; ++i) {
    const int64_t nd = nodes[i];
    const int v = c->var_of_node[nd];
    if (v < 0) continue;
    for (int a = 0; a < 2; ++a) {
      double val = x0[2 * nd + a] + alpha * delta[v + a];
      if (val < lo[v + a]) val = lo[v + a];
      if (val > hi[v + a]) val = hi[v + a];
      x1[2 * nd + a] = val;
    }
  }
}

int pxo_ka_solve_problem(pxo_ka_batch* b, const int32_t* nodes, int nn, const int32_t* edges, int m,
                         const pxo_interp_cfg* cfg, const pxo_loss* loss, double bound,
                         const pxo_lm_options* opt, pxo_lm_summary* sum) {
  return pxo_ka_solve_problem_u(b, nodes, nn, edges, m, NULL, 0, cfg, loss, bound, opt, sum);
}

int pxo_ka_solve_problem_u(pxo_ka_batch* b, const int32_t* nodes, int nn, const int32_t* edges, int m,
                           const int32_t* unary, int nu, const pxo_interp_cfg* cfg, const pxo_loss* loss,
                           double bound, const pxo_lm_options* opt, pxo_lm_summary* sum) {
  ka_ctx c;
  c.b = b; c.cfg = cfg; c.loss = loss; c.edges = edges; c.m = m; c.unary = unary; c.nu = nu;
  int* var_of_node = (int*)malloc(sizeof(int) * b->n_nodes);
  for (int64_t i = 0; i < b->n_nodes; ++i) var_of_node[i] = -1;
  /* will_be_optimized_: endpoints of this problem's edges (featuremetric_keypoint_optimizer.h:198-199) */
  uint8_t* used = (uint8_t*)calloc(b->n_nodes, 1);
  for (int e = 0; e < m; ++e) { used[b->edge_src[edges[e]]] = 1; used[b->edge_dst[edges[e]]] = 1; }
  for (int e = 0; e < nu; ++e) used[b->unary_node[unary[e]]] = 1;
  int n = 0;
  for (int i = 0; i < nn; ++i) {
    const int64_t nd = nodes[i];
    if (used[nd] && b->node_const[nd] != 1) { var_of_node[nd] = n; n += 2; }
  }
  free(used);
  c.var_of_node = var_of_node; c.n = n;
  memset(sum, 0, sizeof(*sum));
  sum->num_unknowns = n; sum->termination = PXO_TERM_NO_CONVERGENCE;
  double* x = b->kp;
  if (n == 0 || m + nu == 0) {
    sum->initial_cost = sum->final_cost = ka_evaluate(&c, x, NULL, NULL);
    sum->termination = PXO_TERM_CONVERGENCE;
    free(var_of_node);
    return 0;
  }
  double* lo = (double*)malloc(sizeof(double) * n); double* hi = (double*)malloc(sizeof(double) * n);
  int feasible = 1;
  for (int i = 0; i < nn; ++i) {
    const int64_t nd = nodes[i];
    const int v = var_of_node[nd];
    if (v < 0) continue;
    ka_bounds(b, nd, bound, lo + v, hi + v);
    if (b->node_const[nd] == 2) {   /* destination outside nodes_in_problem: ParameterizeKeypoints skips it (keypoint_optimizer.h:117) */
      lo[v] = lo[v + 1] = -INFINITY; hi[v] = hi[v + 1] = INFINITY;
    }
    for (int a = 0; a < 2; ++a) if (x[2 * nd + a] < lo[v + a] || x[2 * nd + a] > hi[v + a]) feasible = 0;
  }
  double* H = (double*)malloc(sizeof(double) * (size_t)n * n); double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* g = (double*)malloc(sizeof(double) * n); double* gun = (double*)malloc(sizeof(double) * n);
  double* scale = (double*)malloc(sizeof(double) * n);
  double* step = (double*)malloc(sizeof(double) * n); double* delta = (double*)malloc(sizeof(double) * n);
  double* diag = (double*)malloc(sizeof(double) * n);
  double* x1 = (double*)malloc(sizeof(double) * 2 * b->n_nodes);
  double cost = ka_evaluate(&c, x, H, g);
  sum->initial_cost = cost;
  if (!feasible) { /* [upstream] Program::IsFeasible fails -> Solve returns FAILURE, parameters untouched */
    sum->final_cost = cost; sum->termination = PXO_TERM_FAILURE;
    goto done;
  }
  memcpy(gun, g, sizeof(double) * n);
  for (int j = 0; j < n; ++j) scale[j] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(H[(size_t)j * n + j])) : 1.0;
#define KA_SCALE()                                                                       \
  do {                                                                                   \
    for (int a = 0; a < n; ++a) {                                                        \
      g[a] *= scale[a];                                                                  \
      for (int bb = 0; bb < n; ++bb) H[(size_t)a * n + bb] *= scale[a] * scale[bb];     \
    }                                                                                    \
  } while (0)
  KA_SCALE();
  {
    double radius = opt->initial_radius, decrease_factor = 2.0;
    int invalid = 0, reuse_diag = 0;
    while (1) {
      if (sum->iterations >= opt->max_iterations) { sum->termination = PXO_TERM_NO_CONVERGENCE; break; }
      if (radius < opt->min_radius) { sum->termination = PXO_TERM_CONVERGENCE; break; }
      ++sum->iterations;
      if (!reuse_diag)
        for (int j = 0; j < n; ++j) {
          double d = H[(size_t)j * n + j];
          diag[j] = d < opt->min_lm_diagonal ? opt->min_lm_diagonal : (d > opt->max_lm_diagonal ? opt->max_lm_diagonal : d);
        }
      memcpy(A, H, sizeof(double) * (size_t)n * n);
      for (int j = 0; j < n; ++j) { A[(size_t)j * n + j] += diag[j] / radius; step[j] = -g[j]; }
      int ok = chol_solve(n, A, step) == 0;
      double model_cost_change = 0;
      if (ok) {
        double dg = 0, dHd = 0;
        for (int a = 0; a < n; ++a) {
          dg += step[a] * g[a];
          double hr = 0;
          for (int bb = 0; bb < n; ++bb) hr += H[(size_t)a * n + bb] * step[bb];
          dHd += step[a] * hr;
          if (!isfinite(step[a])) ok = 0;
        }
        model_cost_change = -dg - 0.5 * dHd;
        if (!(model_cost_change > 0.0)) ok = 0;
      }
      if (!ok) {
        if (++invalid >= opt->max_consecutive_invalid_steps) { sum->termination = PXO_TERM_FAILURE; break; }
        radius *= 0.5; reuse_diag = 1;
        continue;
      }
      invalid = 0;
      for (int j = 0; j < n; ++j) delta[j] = step[j] * scale[j];
      /* DoLineSearch [upstream trust_region_minimizer.cc]: projected Armijo search along delta,
       * phi(a) = cost(P(x + a delta)), sufficient decrease 1e-4, cubic interpolation, contraction
       * within [1e-3, 0.6], at most 20 evaluations, minimum step 1e-9. */
      {
        double g0 = 0;
        for (int j = 0; j < n; ++j) g0 += gun[j] * delta[j];
        double xc = 1.0, fc, xp = 0, fp = 0;
        int have_prev = 0, iters = 0, success = 0;
        ka_plus(&c, nodes, nn, x, delta, xc, lo, hi, x1);
        fc = ka_evaluate(&c, x1, NULL, NULL);
        while (1) {
          if (isfinite(fc) && fc <= cost + 1e-4 * g0 * xc) { success = 1; break; }
          if (++iters >= 20) break;
          double nx;
          if (!isfinite(fc)) nx = fmin(fmax(xc * 0.5, xc * 1e-3), xc * 0.6);
          else nx = interpolating_step(cost, g0, have_prev, xp, fp, xc, fc, xc * 1e-3, xc * 0.6);
          if (nx < 1e-9) break;
          if (isfinite(fc)) { xp = xc; fp = fc; have_prev = 1; }
          xc = nx;
          ka_plus(&c, nodes, nn, x, delta, xc, lo, hi, x1);
          fc = ka_evaluate(&c, x1, NULL, NULL);
        }
        if (success) for (int j = 0; j < n; ++j) delta[j] *= xc;
      }
      ka_plus(&c, nodes, nn, x, delta, 1.0, lo, hi, x1);
      const double cand = ka_evaluate(&c, x1, NULL, NULL);
      double step2 = 0, x2 = 0;
      for (int i = 0; i < nn; ++i) {
        const int64_t nd = nodes[i];
        if (var_of_node[nd] < 0) continue;
        for (int a = 0; a < 2; ++a) { const double d = x1[2 * nd + a] - x[2 * nd + a]; step2 += d * d; x2 += x[2 * nd + a] * x[2 * nd + a]; }
      }
      if (sqrt(step2) <= opt->parameter_tolerance * (sqrt(x2) + opt->parameter_tolerance)) { sum->termination = PXO_TERM_CONVERGENCE; break; }
      const double cost_change = cost - cand;
      if (fabs(cost_change) <= opt->function_tolerance * cost) { sum->termination = PXO_TERM_CONVERGENCE; break; }
      const double rel = cost_change / model_cost_change;
      if (rel > opt->min_relative_decrease) {
        for (int i = 0; i < nn; ++i) { const int64_t nd = nodes[i]; x[2 * nd] = x1[2 * nd]; x[2 * nd + 1] = x1[2 * nd + 1]; }
        cost = ka_evaluate(&c, x, H, g);
        memcpy(gun, g, sizeof(double) * n);
        KA_SCALE();
        ++sum->num_successful;
        const double tmp = 2.0 * rel - 1.0;
        double f = 1.0 - tmp * tmp * tmp;
        if (f < 1.0 / 3.0) f = 1.0 / 3.0;
        radius = radius / f;
        if (radius > opt->max_radius) radius = opt->max_radius;
        decrease_factor = 2.0; reuse_diag = 0;
      } else {
        radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = 1;
      }
    }
    sum->final_radius = radius;
  }
  sum->final_cost = cost;
done:
  free(lo); free(hi); free(H); free(A); free(g); free(gun); free(scale); free(step); free(delta); free(diag); free(x1);
  free(var_of_node);
  return 0;
}

/*
 * pxo_lm_bench.c -- CPU ORACLE (TEST INFRASTRUCTURE): ONE Levenberg-Marquardt iteration of the featuremetric BA the way
 * the reference's CPU path spends it, for bench.py's `cpu_baseline_lm` leg and as an independent check of the GPU
 * solver's first step.  What Ceres does per iteration with DENSE_SCHUR / SPARSE_SCHUR
 * (bundle_adjustment/src/bundle_optimizer.h:180-191, num_threads = all cores, bundle_adjustment_options.h:58):
 *   1. residuals + materialised 128 x (10 + K) Jacobians of every residual block (AutoDiffCostFunction), loss +
 *      corrector [upstream corrector.cc];
 *   2. Schur elimination of the point blocks into the reduced camera system [upstream schur_eliminator_impl.h]:
 *      per point E^T E, E^T F_i, then S -= F_i^T E (E^T E + D)^-1 E^T F_j over its observation pairs;
 *   3. Cholesky of the reduced camera system and the camera step;
 *   4. back-substitution of the points;
 *   5. residual-only evaluation at the candidate point.
 * All five stages are threaded over observations / points with OpenMP (static chunks), the Cholesky is a blocked
 * right-looking factorisation threaded over the trailing update.  No Jacobi scaling (opt->jacobi_scaling = 0 on the
 * GPU side gives the same step).  Plain C, no Eigen / Ceres: kind "port" in bench.py's vocabulary.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "pxo.h"

static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

#define PXO_KPAD 12
#define DCMAX (6 + PXO_KPAD)

typedef struct {          /* per-observation normal-equation blocks (tangent, corrected) */
  double A[DCMAX * DCMAX]; /* F^T F   (dc x dc) */
  double W[DCMAX * 3];     /* F^T E   (dc x 3)  */
  double V[9];             /* E^T E             */
  double gc[DCMAX], gp[3];
  int dc, cols[DCMAX];
} obs_blocks;

static pxo_patch patch_of(const pxo_ba_batch* b, int64_t pi) {
  const size_t es = b->dtype == PXO_F16 ? 2 : (b->dtype == PXO_F32 ? 4 : 8);
  pxo_patch p;
  p.data = (const char*)b->arena + (size_t)pi * b->H * b->W * b->C * es;
  p.dtype = b->dtype; p.H = b->H; p.W = b->W; p.C = b->C;
  p.x0 = b->corners[2 * pi]; p.y0 = b->corners[2 * pi + 1];
  p.sx = b->scales[2 * pi]; p.sy = b->scales[2 * pi + 1]; p.up = b->upsampling > 0.0 ? b->upsampling : 1.0;
  return p;
}

static int inv3(const double* a, double* o) {   /* symmetric 3x3, row-major */
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  if (!(det > 0.0)) return -1;
  const double id = 1.0 / det;
  o[0] = c00 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  o[3] = c01 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  o[6] = c02 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
  return 0;
}

/* blocked right-looking Cholesky (lower) of the n x n row-major SPD matrix A.  The trailing update reads the
 * panel through a transposed copy so that its inner loop runs over contiguous columns (vectorisable without
 * re-association); rows of the trailing matrix are shared out over at most 32 threads once more than 3000 of them
 * are left (below that one core is faster than waking a team: 1593 unknowns factor in ~0.15 s serially). */
static int chol_blocked(int n, double* A) {
  const int NB = 64;
  double* P = (double*)malloc(sizeof(double) * (size_t)NB * (n > 0 ? n : 1));
  int nt = omp_get_max_threads();
  if (nt > 32) nt = 32;
  int info = 0;
  for (int k = 0; k < n && !info; k += NB) {
    const int kb = (n - k) < NB ? (n - k) : NB;
    for (int j = k; j < k + kb && !info; ++j) {   /* diagonal block, serial */
      double d = A[(size_t)j * n + j];
      for (int t = k; t < j; ++t) d -= A[(size_t)j * n + t] * A[(size_t)j * n + t];
      if (!(d > 0.0)) { info = j + 1; break; }
      d = sqrt(d);
      A[(size_t)j * n + j] = d;
      for (int i = j + 1; i < k + kb; ++i) {
        double v = A[(size_t)i * n + j];
        for (int t = k; t < j; ++t) v -= A[(size_t)i * n + t] * A[(size_t)j * n + t];
        A[(size_t)i * n + j] = v / d;
      }
    }
    if (info) break;
    const int r0 = k + kb;
#pragma omp parallel for schedule(static) num_threads(nt) if (n - r0 > 3000)
    for (int i = r0; i < n; ++i) {                /* panel: rows below solve against the diagonal block */
      for (int j = k; j < k + kb; ++j) {
        double v = A[(size_t)i * n + j];
        for (int t = k; t < j; ++t) v -= A[(size_t)i * n + t] * A[(size_t)j * n + t];
        A[(size_t)i * n + j] = v / A[(size_t)j * n + j];
      }
      for (int t = 0; t < kb; ++t) P[(size_t)t * n + i] = A[(size_t)i * n + k + t];
    }
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt) if (n - r0 > 3000)
    for (int i = r0; i < n; ++i) {                /* trailing update, lower triangle: A[i][r0..i] -= sum_t L[i][t] L[.][t] */
      double* Ai = A + (size_t)i * n;
      for (int t = 0; t < kb; ++t) {
        const double a = Ai[k + t];
        const double* Pt = P + (size_t)t * n;
#pragma omp simd
        for (int j = r0; j <= i; ++j) Ai[j] -= a * Pt[j];
      }
    }
  }
  free(P);
  return info;
}

/* times_ms: [0] Jacobian evaluation (+ per-block products), [1] Schur elimination, [2] Cholesky + camera step,
 * [3] back-substitution, [4] residual-only evaluation (what the candidate point costs), [5] total.
 * delta_c_out [n_c] / delta_p_out [n_points][3]: the (unscaled, tangent) LM step, may be NULL.
 * Returns 0, or a positive pivot index if the reduced system is not positive definite. */
int pxo_ba_lm_iteration_schur(const pxo_ba_batch* b, int n_images, int n_cams, int64_t n_points,
                              const pxo_interp_cfg* cfg, const pxo_loss* loss, const uint8_t* pose_const,
                              const uint8_t* tvec_const_mask, const uint16_t* cam_const_mask,
                              const uint8_t* point_const, double radius, double min_diag, double max_diag,
                              int n_threads, int* n_c_out, double* delta_c_out, double* delta_p_out,
                              double* times_ms, double* cost_out) {
  if (n_threads > 0) omp_set_num_threads(n_threads);
  const int C = b->C;
  const int64_t n_obs = b->n_obs;
  int* pose_off = (int*)malloc(sizeof(int) * n_images); int* pose_dim = (int*)malloc(sizeof(int) * n_images);
  int* intr_off = (int*)malloc(sizeof(int) * n_cams); int* intr_dim = (int*)malloc(sizeof(int) * n_cams);
  int off = 0;
  for (int i = 0; i < n_images; ++i) {
    int d = 0;
    if (!pose_const[i]) { d = 3; for (int k = 0; k < 3; ++k) if (!((tvec_const_mask[i] >> k) & 1)) ++d; }
    pose_off[i] = off; pose_dim[i] = d; off += d;
  }
  for (int j = 0; j < n_cams; ++j) {
    const int K = pxo_camera_num_params(b->cam_model[j]);
    int d = 0;
    for (int k = 0; k < K; ++k) if (!((cam_const_mask[j] >> k) & 1)) ++d;
    intr_off[j] = off; intr_dim[j] = d; off += d;
  }
  const int n_c = off;
  if (n_c_out) *n_c_out = n_c;
  obs_blocks* ob = (obs_blocks*)malloc(sizeof(obs_blocks) * (size_t)n_obs);
  /* observations of each point (CSR) */
  int64_t* pt_ptr = (int64_t*)calloc(n_points + 1, sizeof(int64_t));
  for (int64_t i = 0; i < n_obs; ++i) ++pt_ptr[b->obs_point[i] + 1];
  for (int64_t p = 0; p < n_points; ++p) pt_ptr[p + 1] += pt_ptr[p];
  int64_t* pt_obs = (int64_t*)malloc(sizeof(int64_t) * (size_t)n_obs);
  { int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * n_points); memcpy(cur, pt_ptr, sizeof(int64_t) * n_points);
    for (int64_t i = 0; i < n_obs; ++i) pt_obs[cur[b->obs_point[i]]++] = i;
    free(cur); }

  const double t0 = now_ms();
  double cost = 0.0;
  /* ---- 1. residuals + Jacobians + corrector, per-block products ---------------------------------------- */
#pragma omp parallel reduction(+ : cost)
  {
    double* r = (double*)malloc(sizeof(double) * C);
    double* Jq = (double*)malloc(sizeof(double) * C * (4 + 3 + 3 + PXO_KPAD));
    double* Jt = Jq + C * 4; double* JX = Jt + C * 3; double* Jk = JX + C * 3;
    double* J = (double*)malloc(sizeof(double) * C * (DCMAX + 3));
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n_obs; ++i) {
      const int img = b->obs_image[i], pt = b->obs_point[i], cam = b->image_camera[img];
      const pxo_patch p = patch_of(b, b->obs_patch[i]);
      const int model = b->cam_model[cam], K = pxo_camera_num_params(model);
      const double* q = b->qvec + 4 * img;
      pxo_ba_residual(&p, cfg, model, q, b->tvec + 3 * img, b->xyz + 3 * (int64_t)pt, b->cam_params + PXO_KPAD * cam,
                      b->refs ? b->refs + (int64_t)C * pt : NULL, r, Jq, Jt, JX, Jk);
      obs_blocks* o = &ob[i];
      int dc = 0;
      /* tangent columns: QuaternionManifold::PlusJacobian [upstream], free tvec components, free intrinsics */
      const double PJ[4][3] = {{-q[1], -q[2], -q[3]}, {q[0], q[3], -q[2]}, {-q[3], q[0], q[1]}, {q[2], -q[1], q[0]}};
      const int n = pose_dim[img] + intr_dim[cam] + 3;
      if (pose_dim[img] > 0) {
        for (int a = 0; a < 3; ++a, ++dc) {
          o->cols[dc] = pose_off[img] + dc;
          for (int c = 0; c < C; ++c)
            J[c * n + dc] = Jq[c * 4] * PJ[0][a] + Jq[c * 4 + 1] * PJ[1][a] + Jq[c * 4 + 2] * PJ[2][a] + Jq[c * 4 + 3] * PJ[3][a];
        }
        for (int a = 0; a < 3; ++a) {
          if ((tvec_const_mask[img] >> a) & 1) continue;
          o->cols[dc] = pose_off[img] + dc;
          for (int c = 0; c < C; ++c) J[c * n + dc] = Jt[c * 3 + a];
          ++dc;
        }
      }
      if (intr_dim[cam] > 0) {
        int kc = 0;
        for (int a = 0; a < K; ++a) {
          if ((cam_const_mask[cam] >> a) & 1) continue;
          o->cols[dc] = intr_off[cam] + kc;
          for (int c = 0; c < C; ++c) J[c * n + dc] = Jk[c * K + a];
          ++dc; ++kc;
        }
      }
      o->dc = dc;
      const int pvar = !point_const[pt];
      for (int a = 0; a < 3; ++a)
        for (int c = 0; c < C; ++c) J[c * n + dc + a] = pvar ? JX[c * 3 + a] : 0.0;
      double s = 0.0;
      for (int c = 0; c < C; ++c) s += r[c] * r[c];
      double rho[3];
      pxo_loss_eval(loss, 1.0, s, rho);
      cost += 0.5 * rho[0];
      pxo_corrector(s, rho, C, n, r, J);
      /* J^T J and J^T r of the block, split into camera (F) and point (E) parts */
      double JTJ[(DCMAX + 3) * (DCMAX + 3)], JTr[DCMAX + 3];
      memset(JTJ, 0, sizeof(double) * n * n); memset(JTr, 0, sizeof(double) * n);
      for (int c = 0; c < C; ++c) {
        const double* row = J + c * n;
        for (int x = 0; x < n; ++x) {
          JTr[x] += row[x] * r[c];
          for (int y = x; y < n; ++y) JTJ[x * n + y] += row[x] * row[y];
        }
      }
      for (int x = 0; x < dc; ++x) {
        o->gc[x] = JTr[x];
        for (int y = 0; y < dc; ++y) o->A[x * DCMAX + y] = x <= y ? JTJ[x * n + y] : JTJ[y * n + x];
        for (int y = 0; y < 3; ++y) o->W[x * 3 + y] = JTJ[x * n + dc + y];
      }
      for (int x = 0; x < 3; ++x) {
        o->gp[x] = JTr[dc + x];
        for (int y = 0; y < 3; ++y) o->V[x * 3 + y] = x <= y ? JTJ[(dc + x) * n + dc + y] : JTJ[(dc + y) * n + dc + x];
      }
    }
    free(r); free(Jq); free(J);
  }
  const double t1 = now_ms();
  /* ---- 2. Schur elimination ---------------------------------------------------------------------------- */
  const size_t ld = (size_t)n_c;
  int nt = omp_get_max_threads();
  if (nt > 32) nt = 32;                      /* one private copy of S per thread */
  double* Sall = (double*)calloc((size_t)nt * (ld * ld + 2 * ld), sizeof(double));
  double* Tall = (double*)malloc(sizeof(double) * 9 * (size_t)n_points);
  double* gpall = (double*)calloc(3 * (size_t)n_points, sizeof(double));
  int bad = 0;
#pragma omp parallel num_threads(nt)
  {
    double* S = Sall + (size_t)omp_get_thread_num() * (ld * ld + 2 * ld);
    double* rhs = S + ld * ld; double* diagU = rhs + ld;
#pragma omp for schedule(static)
    for (int64_t p = 0; p < n_points; ++p) {
      double V[9] = {0}, g[3] = {0};
      for (int64_t e = pt_ptr[p]; e < pt_ptr[p + 1]; ++e) {
        const obs_blocks* o = &ob[pt_obs[e]];
        for (int x = 0; x < 9; ++x) V[x] += o->V[x];
        for (int x = 0; x < 3; ++x) g[x] += o->gp[x];
        for (int x = 0; x < o->dc; ++x) {
          rhs[o->cols[x]] += o->gc[x];
          diagU[o->cols[x]] += o->A[x * DCMAX + x];
          for (int y = 0; y < o->dc; ++y) S[(size_t)o->cols[x] * ld + o->cols[y]] += o->A[x * DCMAX + y];
        }
      }
      double* T = Tall + 9 * p;
      memset(T, 0, sizeof(double) * 9);
      if (point_const[p] || pt_ptr[p] == pt_ptr[p + 1]) continue;
      for (int x = 0; x < 3; ++x) {
        double d = V[x * 4];
        d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d);
        V[x * 4] += d / radius;
      }
      if (inv3(V, T)) { bad = 1; continue; }
      for (int x = 0; x < 3; ++x) gpall[3 * p + x] = g[x];
      for (int64_t e = pt_ptr[p]; e < pt_ptr[p + 1]; ++e) {
        const obs_blocks* oi = &ob[pt_obs[e]];
        double Y[DCMAX * 3];
        for (int x = 0; x < oi->dc; ++x)
          for (int y = 0; y < 3; ++y)
            Y[x * 3 + y] = oi->W[x * 3] * T[y] + oi->W[x * 3 + 1] * T[3 + y] + oi->W[x * 3 + 2] * T[6 + y];
        for (int x = 0; x < oi->dc; ++x)
          rhs[oi->cols[x]] -= Y[x * 3] * g[0] + Y[x * 3 + 1] * g[1] + Y[x * 3 + 2] * g[2];
        for (int64_t f = pt_ptr[p]; f < pt_ptr[p + 1]; ++f) {
          const obs_blocks* oj = &ob[pt_obs[f]];
          for (int x = 0; x < oi->dc; ++x)
            for (int y = 0; y < oj->dc; ++y)
              S[(size_t)oi->cols[x] * ld + oj->cols[y]] -=
                  Y[x * 3] * oj->W[y * 3] + Y[x * 3 + 1] * oj->W[y * 3 + 1] + Y[x * 3 + 2] * oj->W[y * 3 + 2];
        }
      }
    }
  }
  double* S = Sall; double* rhs = S + ld * ld; double* diagU = rhs + ld;
#pragma omp parallel for schedule(static)
  for (int64_t x = 0; x < (int64_t)(ld * ld + 2 * ld); ++x)
    for (int t = 1; t < nt; ++t) S[x] += Sall[(size_t)t * (ld * ld + 2 * ld) + x];
  for (int x = 0; x < n_c; ++x) {
    double d = diagU[x];
    d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d);
    S[(size_t)x * ld + x] += d / radius;
  }
  const double t2 = now_ms();
  /* ---- 3. Cholesky + camera step -------------------------------------------------------------------------- */
  int info = n_c > 0 ? chol_blocked(n_c, S) : 0;
  double* dc_ = (double*)calloc(n_c ? n_c : 1, sizeof(double));
  if (!info) {
    for (int i = 0; i < n_c; ++i) {           /* L y = -rhs */
      double v = -rhs[i];
      for (int t = 0; t < i; ++t) v -= S[(size_t)i * ld + t] * dc_[t];
      dc_[i] = v / S[(size_t)i * ld + i];
    }
    for (int i = n_c - 1; i >= 0; --i) {      /* L^T x = y, row-oriented (contiguous reads of row i of L) */
      const double xi = dc_[i] / S[(size_t)i * ld + i];
      dc_[i] = xi;
      for (int t = 0; t < i; ++t) dc_[t] -= S[(size_t)i * ld + t] * xi;
    }
  }
  const double t3 = now_ms();
  /* ---- 4. back-substitution ---------------------------------------------------------------------------------- */
#pragma omp parallel for schedule(static)
  for (int64_t p = 0; p < n_points; ++p) {
    double v[3] = {gpall[3 * p], gpall[3 * p + 1], gpall[3 * p + 2]};
    for (int64_t e = pt_ptr[p]; e < pt_ptr[p + 1]; ++e) {
      const obs_blocks* o = &ob[pt_obs[e]];
      for (int x = 0; x < o->dc; ++x)
        for (int y = 0; y < 3; ++y) v[y] += o->W[x * 3 + y] * dc_[o->cols[x]];
    }
    const double* T = Tall + 9 * p;
    if (delta_p_out)
      for (int y = 0; y < 3; ++y) delta_p_out[3 * p + y] = -(T[y * 3] * v[0] + T[y * 3 + 1] * v[1] + T[y * 3 + 2] * v[2]);
  }
  const double t4 = now_ms();
  /* ---- 5. residual-only evaluation (the price of the candidate point) ---------------------------------------- */
  double cost2 = 0.0;
#pragma omp parallel reduction(+ : cost2)
  {
    double* r = (double*)malloc(sizeof(double) * C);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n_obs; ++i) {
      const int img = b->obs_image[i], pt = b->obs_point[i], cam = b->image_camera[img];
      const pxo_patch p = patch_of(b, b->obs_patch[i]);
      pxo_ba_residual(&p, cfg, b->cam_model[cam], b->qvec + 4 * img, b->tvec + 3 * img, b->xyz + 3 * (int64_t)pt,
                      b->cam_params + PXO_KPAD * cam, b->refs ? b->refs + (int64_t)C * pt : NULL, r, NULL, NULL, NULL, NULL);
      double s = 0.0;
      for (int c = 0; c < C; ++c) s += r[c] * r[c];
      double rho[3];
      pxo_loss_eval(loss, 1.0, s, rho);
      cost2 += 0.5 * rho[0];
    }
    free(r);
  }
  const double t5 = now_ms();
  if (delta_c_out) memcpy(delta_c_out, dc_, sizeof(double) * n_c);
  if (times_ms) {
    times_ms[0] = t1 - t0; times_ms[1] = t2 - t1; times_ms[2] = t3 - t2; times_ms[3] = t4 - t3; times_ms[4] = t5 - t4;
    times_ms[5] = t5 - t0;
  }
  if (cost_out) { cost_out[0] = cost; cost_out[1] = cost2; }
  free(pose_off); free(pose_dim); free(intr_off); free(intr_dim); free(ob); free(pt_ptr); free(pt_obs);
  free(Sall); free(Tall); free(gpall); free(dc_);
  return bad ? -1 : info;
}

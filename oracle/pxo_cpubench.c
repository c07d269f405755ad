/* pxo_cpubench.c -- ORACLE / TEST INFRASTRUCTURE ONLY: the CPU legs of bench.py (`cpu_baseline`, kind "port").
 *
 * Times the oracle's restatement of the reference's CPU path with the harness of pxo_bench_harness.h (persistent pinned
 * threads, thread-local first-touched copies of the sample, clock inside C):
 *   pxo_bench_ba_eval   one residual block = FeatureReferenceCostFunctor through ceres::AutoDiffCostFunction
 *                       (pixsfm/residuals/src/feature_reference.h:98-137): residual + the materialised 128 x (10+K)
 *                       Jacobian + loss, threaded over residual blocks like Ceres' num_threads = -1
 *                       (pixsfm/bundle_adjustment/src/bundle_adjustment_options.h:58);
 *   pxo_bench_ka_solve  one task = one independent keypoint-adjustment sub-problem solved single-threaded, the tasks
 *                       dealt to a thread pool (pixsfm/keypoint_adjustment/main.py:66-80, base/src/parallel_optimizer.h:77-211).
 */
#define _GNU_SOURCE        /* CPU_SET, pthread barriers / affinity under -std=c11 */
#include <stdatomic.h>
#include <stdio.h>

#include "pxo.h"
#include "pxo_bench_harness.h"

#define PXO_MAXC 512      /* as in pxo_geom.c */
#define PXO_KPAD 12

/* ---- BA residual blocks ---------------------------------------------------------------------------------------- */
typedef struct {
  const pxo_ba_batch* b; const pxo_interp_cfg* cfg; const pxo_loss* loss;
  int64_t n;                 /* observations of the sample */
  int local_copies;          /* 1: every thread copies its share of the arena (first touch on its own CPU) */
  double cost_sink;          /* keeps the work observable */
} ba_bench_user;

typedef struct {
  pxo_ba_batch b;            /* the thread's share: obs arrays offset, arena / corners / scales local */
  int64_t count;
  void* arena; int32_t* corners; double* scales; int64_t* patch;
  double* J;                 /* [C][10 + KPAD] */
  double* Jq;
  double cost;
} ba_bench_state;

static void* ba_bench_init(void* userp, int t, int T) {
  ba_bench_user* u = (ba_bench_user*)userp;
  ba_bench_state* s = (ba_bench_state*)calloc(1, sizeof(ba_bench_state));
  int64_t first, count;
  pxo_bench_share(u->n, t, T, &first, &count);
  const pxo_ba_batch* g = u->b;
  s->b = *g;
  s->count = count;
  s->b.n_obs = count;
  s->b.obs_image = g->obs_image + first; s->b.obs_point = g->obs_point + first;
  const size_t es = g->dtype == PXO_F16 ? 2 : (g->dtype == PXO_F32 ? 4 : 8);
  const size_t pb = (size_t)g->H * g->W * g->C * es;
  if (u->local_copies && count > 0) {
    s->arena = malloc(pb * (size_t)count);
    s->corners = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)count);
    s->scales = (double*)malloc(sizeof(double) * 2 * (size_t)count);
    s->patch = (int64_t*)malloc(sizeof(int64_t) * (size_t)count);
    for (int64_t i = 0; i < count; ++i) {
      const int64_t pi = g->obs_patch[first + i];
      memcpy((char*)s->arena + pb * (size_t)i, (const char*)g->arena + pb * (size_t)pi, pb);
      s->corners[2 * i] = g->corners[2 * pi]; s->corners[2 * i + 1] = g->corners[2 * pi + 1];
      s->scales[2 * i] = g->scales[2 * pi]; s->scales[2 * i + 1] = g->scales[2 * pi + 1];
      s->patch[i] = i;
    }
    s->b.arena = s->arena; s->b.corners = s->corners; s->b.scales = s->scales; s->b.obs_patch = s->patch;
  } else {
    s->b.obs_patch = g->obs_patch + first;
  }
  s->Jq = (double*)malloc(sizeof(double) * g->C * (4 + 3 + 3 + PXO_KPAD));
  s->J = (double*)malloc(sizeof(double) * g->C * (10 + PXO_KPAD));
  return s;
}

static void ba_bench_work(void* userp, void* statep, int t, int T) {
  (void)t; (void)T;
  ba_bench_user* u = (ba_bench_user*)userp;
  ba_bench_state* s = (ba_bench_state*)statep;
  const pxo_ba_batch* b = &s->b;
  const int C = b->C, NJ = 10 + PXO_KPAD;
  const size_t es = b->dtype == PXO_F16 ? 2 : (b->dtype == PXO_F32 ? 4 : 8);
  double r[PXO_MAXC];
  double* Jq = s->Jq; double* Jt = Jq + C * 4; double* JX = Jt + C * 3; double* Jk = JX + C * 3;
  double cost = 0;
  for (int64_t i = 0; i < s->count; ++i) {
    const int img = b->obs_image[i], pt = b->obs_point[i], cam = b->image_camera[img];
    const int64_t pi = b->obs_patch[i];
    pxo_patch p;
    p.data = (const char*)b->arena + (size_t)pi * b->H * b->W * C * es;
    p.dtype = b->dtype; p.H = b->H; p.W = b->W; p.C = C;
    p.x0 = b->corners[2 * pi]; p.y0 = b->corners[2 * pi + 1];
    p.sx = b->scales[2 * pi]; p.sy = b->scales[2 * pi + 1]; p.up = b->upsampling > 0.0 ? b->upsampling : 1.0;
    const int model = b->cam_model[cam];
    const int K = pxo_camera_num_params(model);
    pxo_ba_residual(&p, u->cfg, model, b->qvec + 4 * img, b->tvec + 3 * img, b->xyz + 3 * (int64_t)pt,
                    b->cam_params + PXO_KPAD * cam, b->refs ? b->refs + (int64_t)C * pt : NULL, r, Jq, Jt, JX, Jk);
    double sq = 0;
    for (int c = 0; c < C; ++c) sq += r[c] * r[c];
    double rho[3];
    pxo_loss_eval(u->loss, 1.0, sq, rho);
    cost += 0.5 * rho[0];
    /* the row-major C x (10 + K) block Ceres is handed */
    for (int c = 0; c < C; ++c) {
      double* row = s->J + c * NJ;
      memcpy(row, Jq + c * 4, 32); memcpy(row + 4, Jt + c * 3, 24); memcpy(row + 7, JX + c * 3, 24);
      memcpy(row + 10, Jk + c * K, sizeof(double) * K);
    }
  }
  s->cost = cost;
}

static void ba_bench_fini(void* userp, void* statep) {
  ba_bench_user* u = (ba_bench_user*)userp;
  ba_bench_state* s = (ba_bench_state*)statep;
  /* (racy add of a sink value: only its existence matters) */
  u->cost_sink += s->cost + s->J[0];
  free(s->arena); free(s->corners); free(s->scales); free(s->patch); free(s->Jq); free(s->J); free(s);
}

/* Residual blocks of observations [0, n) of the batch on n_threads persistent threads; out[0] = seconds of the timed
 * region, out[1] = passes over the n blocks inside it, out[2] = seconds of the calibration pass, out[3] = pinned. */
int pxo_bench_ba_eval(const pxo_ba_batch* b, const pxo_interp_cfg* cfg, const pxo_loss* loss, int64_t n, int n_threads,
                      double min_seconds, int local_copies, double* out) {
  ba_bench_user u;
  u.b = b; u.cfg = cfg; u.loss = loss; u.n = n; u.local_copies = local_copies; u.cost_sink = 0;
  pxo_bench_ops ops = {ba_bench_init, ba_bench_work, ba_bench_fini, NULL};
  pxo_bench_result r;
  const int rc = pxo_bench_run(&ops, &u, n_threads, min_seconds, 0, &r);
  if (rc) return rc;
  out[0] = r.seconds; out[1] = (double)r.passes; out[2] = r.calib_seconds; out[3] = r.pinned;
  return 0;
}

/* ---- KA sub-problems -------------------------------------------------------------------------------------------- */
typedef struct {
  const pxo_ka_batch* b;            /* kp is NOT written: every solve works on a private copy of the keypoints */
  int n_problems;
  const int64_t* node_ptr; const int32_t* nodes;     /* CSR: nodes of sub-problem p */
  const int64_t* edge_ptr; const int32_t* edges;
  const pxo_interp_cfg* cfg; const pxo_loss* loss; double bound; const pxo_lm_options* opt;
  atomic_int next;
  atomic_llong iterations;
} ka_bench_user;

typedef struct { double* kp; } ka_bench_state;

static void* ka_bench_init(void* userp, int t, int T) {
  (void)t; (void)T;
  ka_bench_user* u = (ka_bench_user*)userp;
  ka_bench_state* s = (ka_bench_state*)calloc(1, sizeof(ka_bench_state));
  s->kp = (double*)malloc(sizeof(double) * 2 * (size_t)u->b->n_nodes);
  return s;
}
static void ka_bench_work(void* userp, void* statep, int t, int T) {
  (void)t; (void)T;
  ka_bench_user* u = (ka_bench_user*)userp;
  ka_bench_state* s = (ka_bench_state*)statep;
  pxo_ka_batch b = *u->b;
  b.kp = s->kp;
  long long its = 0;
  for (;;) {                         /* a thread pool: the next free thread takes the next sub-problem */
    const int p = atomic_fetch_add(&u->next, 1);
    if (p >= u->n_problems) break;
    const int32_t* nodes = u->nodes + u->node_ptr[p];
    const int nn = (int)(u->node_ptr[p + 1] - u->node_ptr[p]);
    for (int k = 0; k < nn; ++k) {   /* the solve refines in place: start every pass from the initial keypoints */
      s->kp[2 * nodes[k]] = u->b->kp[2 * nodes[k]]; s->kp[2 * nodes[k] + 1] = u->b->kp[2 * nodes[k] + 1];
    }
    pxo_lm_summary sum;
    pxo_ka_solve_problem(&b, nodes, nn, u->edges + u->edge_ptr[p], (int)(u->edge_ptr[p + 1] - u->edge_ptr[p]), u->cfg, u->loss,
                         u->bound, u->opt, &sum);
    its += sum.iterations;
  }
  atomic_fetch_add(&u->iterations, its);
}
static void ka_bench_fini(void* userp, void* statep) {
  (void)userp;
  ka_bench_state* s = (ka_bench_state*)statep;
  free(s->kp); free(s);
}
static void ka_bench_between(void* userp) {
  ka_bench_user* u = (ka_bench_user*)userp;
  atomic_store(&u->next, 0);
}

/* All n_problems sub-problems once per pass, dealt dynamically to n_threads threads (each solve single-threaded).
 * out[0] = seconds, out[1] = passes, out[2] = calibration seconds, out[3] = pinned, out[4] = LM iterations summed over all
 * solves of all passes (calibration included). */
int pxo_bench_ka_solve(const pxo_ka_batch* b, int n_problems, const int64_t* node_ptr, const int32_t* nodes,
                       const int64_t* edge_ptr, const int32_t* edges, const pxo_interp_cfg* cfg, const pxo_loss* loss,
                       double bound, const pxo_lm_options* opt, int n_threads, double min_seconds, double* out) {
  ka_bench_user u;
  u.b = b; u.n_problems = n_problems; u.node_ptr = node_ptr; u.nodes = nodes; u.edge_ptr = edge_ptr; u.edges = edges;
  u.cfg = cfg; u.loss = loss; u.bound = bound; u.opt = opt;
  atomic_init(&u.next, 0); atomic_init(&u.iterations, 0);
  pxo_bench_ops ops = {ka_bench_init, ka_bench_work, ka_bench_fini, ka_bench_between};
  pxo_bench_result r;
  const int rc = pxo_bench_run(&ops, &u, n_threads, min_seconds, 0, &r);
  if (rc) return rc;
  out[0] = r.seconds; out[1] = (double)r.passes; out[2] = r.calib_seconds; out[3] = r.pinned;
  out[4] = (double)atomic_load(&u.iterations);
  return 0;
}

/* ---- what the host gives N threads: a pure-ALU probe ---------------------------------------------------------------------
 * Every thread runs the same fixed number of dependent fused multiply-adds on registers (no memory traffic).  The rate at
 * N threads over the rate at one thread is the number of cores' worth of arithmetic the box really hands this process --
 * a container on a shared node, a CPU quota or SMT siblings show up here, separately from a leg's own memory behaviour. */
typedef struct { int64_t iters; double sink; } spin_user;
static void spin_work(void* userp, void* statep, int t, int T) {
  (void)statep; (void)T;
  spin_user* u = (spin_user*)userp;
  double a0 = 1.0 + t, a1 = 1.1, a2 = 1.2, a3 = 1.3, a4 = 1.4, a5 = 1.5, a6 = 1.6, a7 = 1.7;
  const double m = 0.999999, c = 1e-9;
  for (int64_t i = 0; i < u->iters; ++i) {
    a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
    a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
  }
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) u->sink += 1.0;
}
/* out[0] = seconds, out[1] = passes, out[2] = calibration seconds, out[3] = pinned; work per pass = iters x 8 FMAs per thread */
int pxo_bench_spin(int64_t iters, int n_threads, double min_seconds, double* out) {
  spin_user u; u.iters = iters; u.sink = 0;
  pxo_bench_ops ops = {NULL, spin_work, NULL, NULL};
  pxo_bench_result r;
  const int rc = pxo_bench_run(&ops, &u, n_threads, min_seconds, 0, &r);
  if (rc) return rc;
  out[0] = r.seconds; out[1] = (double)r.passes; out[2] = r.calib_seconds; out[3] = r.pinned;
  return 0;
}

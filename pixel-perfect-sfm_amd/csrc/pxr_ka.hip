// pxr_ka.hip -- featuremetric keypoint adjustment on gfx950.
//
// Reference path: FeatureMetricKeypointOptimizer::RunParallel -> one ceres::Problem per problem
// label, one FeatureMetric2DCostFunctor per intra-track edge (2 bicubic interpolations per
// residual block, 128 x 2 x 2 Jacobian), solved by Ceres TR-LM with box bounds on a CPU thread
// (keypoint_adjustment/src/featuremetric_keypoint_optimizer.h:69-202, keypoint_optimizer.h:77-157).
//
// MI355X design: ONE workgroup owns one sub-problem for the whole solve -- no host round trips,
// thousands of sub-problems in flight.  Per LM iteration the workgroup
//   1. evaluates every NODE once (16 lanes per node, 8 channels per lane: normalised descriptor
//      + image-space gradients, 3 KiB per node kept in an L2-resident scratch) instead of twice
//      per EDGE as the reference does -- a track of 10 nodes with its complete match graph reads
//      each 4 KiB stencil once, not 18 times;
//   2. walks the edges (16 lanes per edge): r = f_src - f_dst, the 15 dot products that define
//      the robustified 4x4 normal block, reduced with DPP inside the row, scattered into the
//      dense sub-problem normal matrix;
//   3. runs Ceres' trust-region step: Jacobi scaling, LM damping, in-LDS Cholesky, projected
//      Armijo line search along the step (bounds), step acceptance and radius update.
// [upstream Ceres 2.1] semantics restated as in oracle/pxo_solve.c; the oracle is the parity
// target (real Ceres is not available: parity unpinned w.r.t. the reference binary).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <vector>

#include "pxr_device.h"
#include "pxr_interp.h"
#include "pxr_internal.h"

namespace pxr {

constexpr int KA_NLDS = 112;   // largest sub-problem (unknowns) whose damped matrix lives in LDS

struct KaArgs {
  pxr_ka_view v;
  const void* arena; const int32_t* corners; const double* scales; int H, W;
  int l2_normalize; int float_simd; int lds_n;   // lds_n: largest system kept in LDS (<= KA_NLDS)
  pxr_loss loss; double bound; pxr_lm_options opt;
  // scratch
  double* desc;          // [n_nodes][3][C]: f, df/dx, df/dy
  double* kp_cand;       // [n_nodes][2]
  int* var_of_node;      // [n_nodes]
  uint8_t* used;         // [n_nodes] (zeroed by host)
  double* vec;           // 10 vectors of [2 * n_nodes]: g, gun, scale, diag, step, delta, lo, hi, rhs, tmp
  const int64_t* prob_h_ptr;   // [n_problems + 1] offsets into Hbuf / Abuf
  double* Hbuf; double* Abuf;
  pxr_lm_summary* summaries;   // device [n_problems]
};

template <typename ST, int C, bool WITH_JAC>
__device__ __forceinline__ void ka_eval_node(const KaArgs& a, int64_t node, const double* kp, int sub, bool fsimd) {
  const int64_t pi = a.v.d_node_patch[node];
  const double sx = a.scales[2 * pi], sy = a.scales[2 * pi + 1];
  // FeaturePatch::ToPixelCoordinates, features/src/featurepatch.h:250-255
  const double u = kp[2 * node] * sx - 0.5 - (double)a.corners[2 * pi];
  const double v = kp[2 * node + 1] * sy - 0.5 - (double)a.corners[2 * pi + 1];
  const ST* patch = reinterpret_cast<const ST*>(a.arena) + (size_t)pi * a.H * a.W * C;
  double f[8], fr[8], fc[8];
  if (fsimd) interp8<ST, C / 8, WITH_JAC, true>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f, fr, fc);
  else interp8<ST, C / 8, WITH_JAC, false>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f, fr, fc);
  double* d = a.desc + (size_t)node * 3 * C + sub * 8;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    d[ch] = f[ch];
    if (WITH_JAC) { d[C + ch] = fc[ch] * sx; d[2 * C + ch] = fr[ch] * sy; }
  }
}

__device__ __forceinline__ double lpo_sum(double v, int LPO) { return LPO == 16 ? row16_sum(v) : row8_sum(v); }

// block-wide sum, result broadcast to every thread (256 threads)
__device__ __forceinline__ double block_sum(double v, double* sh4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

struct KaProb {
  int64_t np0, np1, ne0, ne1;
  int n;                 // unknowns
  double *g, *gun, *scale, *diag, *step, *delta, *lo, *hi, *rhs;
  double* Hm;            // n x n (global scratch)
};

// evaluate all nodes of the problem at keypoints `kp`
template <typename ST, int C, bool WITH_JAC>
__device__ void ka_nodes(const KaArgs& a, const KaProb& p, const double* kp, bool fsimd) {
  constexpr int LPO = C / 8, G = 256 / LPO;
  const int grp = threadIdx.x / LPO, sub = threadIdx.x % LPO;
  for (int64_t i = p.np0 + grp; i < p.np1; i += G) {
    const int64_t node = a.v.d_prob_nodes[i];
    if (!a.used[node]) continue;
    ka_eval_node<ST, C, WITH_JAC>(a, node, kp, sub, fsimd);
  }
  __syncthreads();
}

// walk the edges; returns the cost (block-uniform).  WITH_JAC: accumulates Hm and g (unscaled).
template <int C, bool WITH_JAC>
__device__ double ka_edges(const KaArgs& a, const KaProb& p, double* sh4) {
  constexpr int LPO = C / 8, G = 256 / LPO;
  const int grp = threadIdx.x / LPO, sub = threadIdx.x % LPO;
  double cost = 0.0;
  for (int64_t i = p.ne0 + grp; i < p.ne1; i += G) {
    const int e = a.v.d_prob_edges[i];
    const int n1 = a.v.d_edge_src[e], n2 = a.v.d_edge_dst[e];
    const double* d1 = a.desc + (size_t)n1 * 3 * C + sub * 8;
    const double* d2 = a.desc + (size_t)n2 * 3 * C + sub * 8;
    double r[8];
    double s = 0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { r[ch] = d1[ch] - d2[ch]; s = fma(r[ch], r[ch], s); }
    s = lpo_sum(s, LPO);
    double rho[3];
    loss_eval(a.loss.type, a.loss.a, a.v.d_edge_w[e], s, rho);
    if (sub == 0) cost += 0.5 * rho[0];
    if (WITH_JAC) {
      // J = [g1x g1y -g2x -g2y]; 10 entries of J^T J and 4 of J^T r
      double q[14];
#pragma unroll
      for (int k = 0; k < 14; ++k) q[k] = 0.0;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const double a0 = d1[C + ch], a1 = d1[2 * C + ch], a2 = -d2[C + ch], a3 = -d2[2 * C + ch];
        q[0] = fma(a0, a0, q[0]); q[1] = fma(a0, a1, q[1]); q[2] = fma(a0, a2, q[2]); q[3] = fma(a0, a3, q[3]);
        q[4] = fma(a1, a1, q[4]); q[5] = fma(a1, a2, q[5]); q[6] = fma(a1, a3, q[6]);
        q[7] = fma(a2, a2, q[7]); q[8] = fma(a2, a3, q[8]); q[9] = fma(a3, a3, q[9]);
        q[10] = fma(a0, r[ch], q[10]); q[11] = fma(a1, r[ch], q[11]);
        q[12] = fma(a2, r[ch], q[12]); q[13] = fma(a3, r[ch], q[13]);
      }
#pragma unroll
      for (int k = 0; k < 14; ++k) q[k] = lpo_sum(q[k], LPO);
      if (sub == 0) {
        double kappa = 0.0;   // corrector [upstream Ceres corrector.cc]
        if (s != 0.0 && rho[2] > 0.0) {
          const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
          const double alpha = 1.0 - sqrt(D);
          kappa = (2.0 * alpha - alpha * alpha) / s;
        }
        const int v1 = a.var_of_node[n1], v2 = a.var_of_node[n2];
        const int idx[4] = {v1, v1 + 1, v2, v2 + 1};
        const bool var[4] = {v1 >= 0, v1 >= 0, v2 >= 0, v2 >= 0};
        const double b[4] = {q[10], q[11], q[12], q[13]};
        const double m[4][4] = {{q[0], q[1], q[2], q[3]}, {q[1], q[4], q[5], q[6]}, {q[2], q[5], q[7], q[8]}, {q[3], q[6], q[8], q[9]}};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          if (!var[x]) continue;
          atomicAdd(p.g + idx[x], rho[1] * b[x]);
#pragma unroll
          for (int y = 0; y < 4; ++y)
            if (var[y]) atomicAdd(p.Hm + (size_t)idx[x] * p.n + idx[y], rho[1] * (m[x][y] - kappa * b[x] * b[y]));
        }
      }
    }
  }
  return block_sum(cost, sh4);
}

// [upstream Ceres polynomial.cc] minimiser of the interpolating quadratic / cubic on [lo, hi]
__device__ double ka_poly(const double* c, int deg, double x) {
  double v = 0;
  for (int i = deg; i >= 0; --i) v = v * x + c[i];
  return v;
}
__device__ double ka_interp_step(double f0, double g0, bool have_prev, double xp, double fp, double xc, double fc,
                                 double lo, double hi) {
  double c[4] = {f0, g0, 0, 0};
  int deg;
  if (!have_prev) { c[2] = (fc - f0 - g0 * xc) / (xc * xc); deg = 2; }
  else {
    const double rp = (fp - f0 - g0 * xp) / (xp * xp), rc = (fc - f0 - g0 * xc) / (xc * xc);
    c[3] = (rc - rp) / (xc - xp); c[2] = rc - c[3] * xc; deg = 3;
  }
  double best_x = lo, best = ka_poly(c, deg, lo);
  double v = ka_poly(c, deg, hi);
  if (v < best) { best = v; best_x = hi; }
  if (deg == 2) {
    if (c[2] != 0.0) { const double x = -c[1] / (2 * c[2]); if (x > lo && x < hi && ka_poly(c, 2, x) < best) best_x = x; }
  } else {
    const double A = 3 * c[3], B = 2 * c[2], Cc = c[1];
    if (A == 0.0) {
      if (B != 0.0) { const double x = -Cc / B; if (x > lo && x < hi && ka_poly(c, 3, x) < best) best_x = x; }
    } else {
      const double disc = B * B - 4 * A * Cc;
      if (disc >= 0) {
        const double sq = sqrt(disc), x1 = (-B + sq) / (2 * A), x2 = (-B - sq) / (2 * A);
        if (x1 > lo && x1 < hi) { v = ka_poly(c, 3, x1); if (v < best) { best = v; best_x = x1; } }
        if (x2 > lo && x2 < hi) { v = ka_poly(c, 3, x2); if (v < best) { best = v; best_x = x2; } }
      }
    }
  }
  return best_x;
}

// candidate = P(x + alpha * delta) for the problem's variable nodes (ParameterBlock::Plus [upstream])
__device__ void ka_plus(const KaArgs& a, const KaProb& p, double alpha) {
  for (int64_t i = p.np0 + threadIdx.x; i < p.np1; i += blockDim.x) {
    const int64_t node = a.v.d_prob_nodes[i];
    const int v = a.var_of_node[node];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double val = a.v.d_kp[2 * node + c];
      if (v >= 0) val = fmin(fmax(val + alpha * p.delta[v + c], p.lo[v + c]), p.hi[v + c]);
      a.kp_cand[2 * node + c] = val;
    }
  }
  __syncthreads();
}

// in-place Cholesky solve of the n x n row-major lower matrix A (LDS or global), rhs -> solution.
// Returns false on a non-positive pivot (block-uniform).
__device__ bool ka_chol_solve(double* A, int n, double* b) {
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  bool ok = true;
  for (int j = 0; j < n; ++j) {
    const double d = A[(size_t)j * n + j];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    const double piv = d > 0.0 ? sqrt(d) : 1.0, inv = 1.0 / piv;
    __syncthreads();
    if (tid == 0) A[(size_t)j * n + j] = piv;
    for (int i = j + 1 + tid; i < n; i += 256) A[(size_t)i * n + j] *= inv;
    __syncthreads();
    for (int i = j + 1 + ti; i < n; i += 16) {
      const double lij = A[(size_t)i * n + j];
      for (int c = j + 1 + tj; c <= i; c += 16) A[(size_t)i * n + c] -= lij * A[(size_t)c * n + j];
    }
    __syncthreads();
  }
  // forward: L y = b (column oriented)
  for (int j = 0; j < n; ++j) {
    const double yj = b[j] / A[(size_t)j * n + j];
    __syncthreads();
    if (tid == 0) b[j] = yj;
    for (int i = j + 1 + tid; i < n; i += 256) b[i] -= A[(size_t)i * n + j] * yj;
    __syncthreads();
  }
  // backward: L^T x = y
  for (int j = n - 1; j >= 0; --j) {
    const double xj = b[j] / A[(size_t)j * n + j];
    __syncthreads();
    if (tid == 0) b[j] = xj;
    for (int i = tid; i < j; i += 256) b[i] -= A[(size_t)j * n + i] * xj;
    __syncthreads();
  }
  return ok;
}

template <typename ST, int C>
__global__ __launch_bounds__(256) void ka_solve_kernel(const KaArgs a) {
  extern __shared__ double sh_A[];      // lds_n^2 doubles (damped matrix) when the sub-problem fits
  __shared__ double sh4[4];
  __shared__ int sh_n, sh_feasible;
  const int prob = blockIdx.x, tid = threadIdx.x;
  const bool fsimd = a.float_simd != 0;
  KaProb p;
  p.np0 = a.v.d_prob_node_ptr[prob]; p.np1 = a.v.d_prob_node_ptr[prob + 1];
  p.ne0 = a.v.d_prob_edge_ptr[prob]; p.ne1 = a.v.d_prob_edge_ptr[prob + 1];
  const size_t vstride = 2 * (size_t)a.v.n_nodes, vb = 2 * (size_t)p.np0;
  p.g = a.vec + 0 * vstride + vb; p.gun = a.vec + 1 * vstride + vb; p.scale = a.vec + 2 * vstride + vb;
  p.diag = a.vec + 3 * vstride + vb; p.step = a.vec + 4 * vstride + vb; p.delta = a.vec + 5 * vstride + vb;
  p.lo = a.vec + 6 * vstride + vb; p.hi = a.vec + 7 * vstride + vb; p.rhs = a.vec + 8 * vstride + vb;
  p.Hm = a.Hbuf + a.prob_h_ptr[prob];
  pxr_lm_summary sm;
  sm.iterations = 0; sm.num_successful = 0; sm.termination = PXR_TERM_NO_CONVERGENCE;
  sm.num_camera_unknowns = 0; sm.num_point_unknowns = 0; sm.initial_cost = 0; sm.final_cost = 0;
  sm.final_radius = 0; sm.total_ms = 0; sm.setup_ms = 0;

  // will_be_optimized_ (featuremetric_keypoint_optimizer.h:198-199): endpoints of this problem's edges
  for (int64_t i = p.ne0 + tid; i < p.ne1; i += blockDim.x) {
    const int e = a.v.d_prob_edges[i];
    a.used[a.v.d_edge_src[e]] = 1; a.used[a.v.d_edge_dst[e]] = 1;
  }
  __syncthreads();
  if (tid == 0) {   // unknown layout in ascending node order + box bounds (keypoint_optimizer.h:127-152)
    int n = 0, feasible = 1;
    for (int64_t i = p.np0; i < p.np1; ++i) {
      const int64_t node = a.v.d_prob_nodes[i];
      int v = -1;
      if (a.used[node] && !a.v.d_node_const[node]) {
        v = n; n += 2;
        const int64_t pi = a.v.d_node_patch[node];
        const double sx = a.scales[2 * pi], sy = a.scales[2 * pi + 1];
        const double kx = a.v.d_kp[2 * node], ky = a.v.d_kp[2 * node + 1];
        double lx = (a.corners[2 * pi] + 0.5) / sx, ly = (a.corners[2 * pi + 1] + 0.5) / sy;
        double ux = lx + a.W / sx, uy = ly + a.H / sy;
        if (a.bound > 0.0) {
          ux = fmin(kx + a.bound / sx, ux); uy = fmin(ky + a.bound / sy, uy);
          lx = fmax(kx - a.bound / sx, lx); ly = fmax(ky - a.bound / sy, ly);
        }
        p.lo[v] = lx; p.lo[v + 1] = ly; p.hi[v] = ux; p.hi[v + 1] = uy;
        if (kx < lx || kx > ux || ky < ly || ky > uy) feasible = 0;
      }
      a.var_of_node[node] = v;
    }
    sh_n = n; sh_feasible = feasible;
  }
  __syncthreads();
  const int n = sh_n;
  p.n = n;
  sm.num_camera_unknowns = n;
  double* A = (n <= a.lds_n) ? sh_A : (a.Abuf + a.prob_h_ptr[prob]);
  const pxr_lm_options& opt = a.opt;

  auto zero_normal = [&]() {
    for (int e = tid; e < n * n; e += blockDim.x) p.Hm[e] = 0.0;
    for (int e = tid; e < n; e += blockDim.x) p.g[e] = 0.0;
    __syncthreads();
  };
  // evaluate cost + normal equations at the CURRENT keypoints, then scale: H <- S H S, g <- S g
  auto linearize = [&](bool compute_scale) -> double {
    zero_normal();
    ka_nodes<ST, C, true>(a, p, a.v.d_kp, fsimd);
    const double c = ka_edges<C, true>(a, p, sh4);
    __syncthreads();
    for (int e = tid; e < n; e += blockDim.x) {
      p.gun[e] = p.g[e];
      if (compute_scale) p.scale[e] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(p.Hm[(size_t)e * n + e])) : 1.0;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += blockDim.x) p.Hm[e] *= p.scale[e / n] * p.scale[e % n];
    for (int e = tid; e < n; e += blockDim.x) p.g[e] *= p.scale[e];
    __syncthreads();
    return c;
  };
  auto cost_at_candidate = [&]() -> double {
    ka_nodes<ST, C, false>(a, p, a.kp_cand, fsimd);
    return ka_edges<C, false>(a, p, sh4);
  };

  if (n == 0 || p.ne1 == p.ne0) {
    ka_nodes<ST, C, false>(a, p, a.v.d_kp, fsimd);
    const double c = ka_edges<C, false>(a, p, sh4);
    sm.initial_cost = sm.final_cost = c; sm.termination = PXR_TERM_CONVERGENCE;
    if (tid == 0) a.summaries[prob] = sm;
    return;
  }
  double cost = linearize(true);
  sm.initial_cost = cost;
  if (!sh_feasible) {   // [upstream] Program::IsFeasible fails: FAILURE, parameters untouched
    sm.final_cost = cost; sm.termination = PXR_TERM_FAILURE;
    if (tid == 0) a.summaries[prob] = sm;
    return;
  }
  double radius = opt.initial_radius, decrease_factor = 2.0;
  int invalid = 0;
  bool reuse_diag = false;
  while (true) {
    if (sm.iterations >= opt.max_iterations) { sm.termination = PXR_TERM_NO_CONVERGENCE; break; }
    if (radius < opt.min_radius) { sm.termination = PXR_TERM_CONVERGENCE; break; }
    ++sm.iterations;
    if (!reuse_diag)
      for (int e = tid; e < n; e += blockDim.x)
        p.diag[e] = fmin(fmax(p.Hm[(size_t)e * n + e], opt.min_lm_diagonal), opt.max_lm_diagonal);
    __syncthreads();
    for (int e = tid; e < n * n; e += blockDim.x) {
      const int i = e / n, j = e % n;
      A[e] = p.Hm[e] + (i == j ? p.diag[i] / radius : 0.0);
    }
    for (int e = tid; e < n; e += blockDim.x) p.step[e] = -p.g[e];
    __syncthreads();
    bool ok = ka_chol_solve(A, n, p.step);
    // model cost change = -d.g - 0.5 d.H.d
    double part = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
      double hr = 0.0;
      for (int j = 0; j < n; ++j) hr = fma(p.Hm[(size_t)i * n + j], p.step[j], hr);
      part += -p.step[i] * p.g[i] - 0.5 * p.step[i] * hr;
      if (!isfinite(p.step[i])) part = NAN;
    }
    const double model_cost_change = block_sum(part, sh4);
    if (!(model_cost_change > 0.0)) ok = false;
    if (!ok) {
      if (++invalid >= opt.max_consecutive_invalid_steps) { sm.termination = PXR_TERM_FAILURE; break; }
      radius *= 0.5; reuse_diag = true;
      continue;
    }
    invalid = 0;
    double g0p = 0.0;
    for (int e = tid; e < n; e += blockDim.x) {
      const double dl = p.step[e] * p.scale[e];
      p.delta[e] = dl;
      g0p += p.gun[e] * dl;
    }
    const double g0 = block_sum(g0p, sh4);
    // DoLineSearch [upstream trust_region_minimizer.cc]: projected Armijo search along delta
    {
      double xc = 1.0, xp = 0.0, fp = 0.0;
      bool have_prev = false, success = false;
      int iters = 0;
      ka_plus(a, p, xc);
      double fc = cost_at_candidate();
      while (true) {
        if (isfinite(fc) && fc <= cost + 1e-4 * g0 * xc) { success = true; break; }
        if (++iters >= 20) break;
        double nx;
        if (!isfinite(fc)) nx = fmin(fmax(xc * 0.5, xc * 1e-3), xc * 0.6);
        else nx = ka_interp_step(cost, g0, have_prev, xp, fp, xc, fc, xc * 1e-3, xc * 0.6);
        if (nx < 1e-9) break;
        if (isfinite(fc)) { xp = xc; fp = fc; have_prev = true; }
        xc = nx;
        ka_plus(a, p, xc);
        fc = cost_at_candidate();
      }
      if (success && xc != 1.0) {
        for (int e = tid; e < n; e += blockDim.x) p.delta[e] *= xc;
        __syncthreads();
      }
    }
    ka_plus(a, p, 1.0);
    const double cand = cost_at_candidate();
    double s2 = 0.0, x2 = 0.0;
    for (int64_t i = p.np0 + tid; i < p.np1; i += blockDim.x) {
      const int64_t node = a.v.d_prob_nodes[i];
      if (a.var_of_node[node] < 0) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double x0 = a.v.d_kp[2 * node + c], d = a.kp_cand[2 * node + c] - x0;
        s2 += d * d; x2 += x0 * x0;
      }
    }
    const double step_norm = sqrt(block_sum(s2, sh4)), x_norm = sqrt(block_sum(x2, sh4));
    if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) { sm.termination = PXR_TERM_CONVERGENCE; break; }
    const double cost_change = cost - cand;
    if (fabs(cost_change) <= opt.function_tolerance * cost) { sm.termination = PXR_TERM_CONVERGENCE; break; }
    const double rel = cost_change / model_cost_change;
    if (rel > opt.min_relative_decrease) {
      for (int64_t i = p.np0 + tid; i < p.np1; i += blockDim.x) {
        const int64_t node = a.v.d_prob_nodes[i];
        a.v.d_kp[2 * node] = a.kp_cand[2 * node]; a.v.d_kp[2 * node + 1] = a.kp_cand[2 * node + 1];
      }
      __syncthreads();
      cost = linearize(false);
      ++sm.num_successful;
      const double tmp = 2.0 * rel - 1.0;
      radius = fmin(opt.max_radius, radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp));
      decrease_factor = 2.0; reuse_diag = false;
    } else {
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = true;
    }
  }
  sm.final_cost = cost; sm.final_radius = radius;
  if (tid == 0) a.summaries[prob] = sm;
}

// ---- per-edge evaluation (parity checks) -------------------------------------------------------------------
template <typename ST, int C>
__global__ __launch_bounds__(256) void ka_eval_kernel(const KaArgs a, bool fsimd, double* __restrict__ cost,
                                                      double* __restrict__ out_r, double* __restrict__ out_J1,
                                                      double* __restrict__ out_J2) {
  constexpr int LPO = C / 8, G = 256 / LPO;
  const int grp = threadIdx.x / LPO, sub = threadIdx.x % LPO;
  const int64_t e = (int64_t)blockIdx.x * G + grp;
  if (e >= a.v.n_edges) return;
  const int nn[2] = {a.v.d_edge_src[e], a.v.d_edge_dst[e]};
  double f[2][8], gx[2][8], gy[2][8];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int64_t pi = a.v.d_node_patch[nn[k]];
    const double sx = a.scales[2 * pi], sy = a.scales[2 * pi + 1];
    const double u = a.v.d_kp[2 * (size_t)nn[k]] * sx - 0.5 - (double)a.corners[2 * pi];
    const double v = a.v.d_kp[2 * (size_t)nn[k] + 1] * sy - 0.5 - (double)a.corners[2 * pi + 1];
    const ST* patch = reinterpret_cast<const ST*>(a.arena) + (size_t)pi * a.H * a.W * C;
    double fr[8], fc[8];
    if (fsimd) interp8<ST, LPO, true, true>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f[k], fr, fc);
    else interp8<ST, LPO, true, false>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f[k], fr, fc);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { gx[k][ch] = fc[ch] * sx; gy[k][ch] = fr[ch] * sy; }
  }
  double s = 0;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) { const double r = f[0][ch] - f[1][ch]; s = fma(r, r, s); }
  s = lpo_sum(s, LPO);
  double rho[3];
  loss_eval(a.loss.type, a.loss.a, a.v.d_edge_w[e], s, rho);
  if (sub == 0) cost[e] = 0.5 * rho[0];
  if (out_r) {
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const size_t o = (size_t)e * C + sub * 8 + ch;
      out_r[o] = f[0][ch] - f[1][ch];
      if (out_J1) { out_J1[2 * o] = gx[0][ch]; out_J1[2 * o + 1] = gy[0][ch]; }
      if (out_J2) { out_J2[2 * o] = -gx[1][ch]; out_J2[2 * o + 1] = -gy[1][ch]; }
    }
  }
}

static int fill_args(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                     const pxr_loss* loss, KaArgs& a) {
  a.v = *view;
  a.arena = arena->d_data; a.corners = arena->d_corners; a.scales = arena->d_scales;
  a.H = arena->H; a.W = arena->W; a.l2_normalize = cfg->l2_normalize; a.float_simd = cfg->use_float_simd; a.loss = *loss;
  return PXR_OK;
}

template <typename T>
struct KaBuf {
  T* p = nullptr;
  int alloc(size_t n) { return hip_check(hipMalloc((void**)&p, sizeof(T) * (n ? n : 1)), "hipMalloc(KA scratch)"); }
  ~KaBuf() { if (p) (void)hipFree(p); }
};

}  // namespace pxr

extern "C" int pxr_ka_eval(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                           const pxr_loss* loss, double* d_cost, double* d_r, double* d_J1, double* d_J2) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && view && cfg && loss && d_cost, "pxr_ka_eval: NULL argument");
  PXR_REQUIRE(!((d_J1 || d_J2) && !d_r), "pxr_ka_eval: Jacobian outputs require d_r");
  if (view->n_edges == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
  KaArgs a{};
  fill_args(ctx, arena, view, cfg, loss, a);
  const bool fs = cfg->use_float_simd != 0;
#define KA_EVAL_LAUNCH(ST, CC)                                                                                   \
  hipLaunchKernelGGL((ka_eval_kernel<ST, CC>), dim3((unsigned)((view->n_edges + (256 / (CC / 8)) - 1) / (256 / (CC / 8)))), \
                     dim3(256), 0, ctx->stream, a, fs, d_cost, d_r, d_J1, d_J2)
  if (arena->dtype == PXR_F16 && arena->C == 128) KA_EVAL_LAUNCH(_Float16, 128);
  else if (arena->dtype == PXR_F16 && arena->C == 64) KA_EVAL_LAUNCH(_Float16, 64);
  else if (arena->dtype == PXR_F32 && arena->C == 128) KA_EVAL_LAUNCH(float, 128);
  else if (arena->dtype == PXR_F32 && arena->C == 64) KA_EVAL_LAUNCH(float, 64);
  else if (arena->dtype == PXR_F64 && arena->C == 128) KA_EVAL_LAUNCH(double, 128);
  else if (arena->dtype == PXR_F64 && arena->C == 64) KA_EVAL_LAUNCH(double, 64);
  else return set_error(PXR_EUNSUPPORTED, "pxr_ka_eval: CHANNELS=%d not supported (128, 64)", arena->C);
#undef KA_EVAL_LAUNCH
  return hip_check(hipGetLastError(), "ka_eval_kernel launch");
}

extern "C" int pxr_ka_solve(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                            const pxr_loss* loss, double bound, const pxr_lm_options* options,
                            pxr_lm_summary* h_summaries, pxr_lm_summary* total) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && view && cfg && loss && options && total, "pxr_ka_solve: NULL argument");
  PXR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int np = view->n_problems;
  memset(total, 0, sizeof(*total));
  if (np == 0) return PXR_OK;
  // per-problem normal-matrix offsets from the node counts (upper bound 2 * nodes unknowns)
  std::vector<int64_t> node_ptr(np + 1), h_ptr(np + 1, 0);
  PXR_HIP(hipMemcpyAsync(node_ptr.data(), view->d_prob_node_ptr, sizeof(int64_t) * (np + 1), hipMemcpyDeviceToHost, st));
  PXR_HIP(hipStreamSynchronize(st));
  bool need_aglob = false;
  int64_t largest = 1;
  for (int i = 0; i < np; ++i) {
    const int64_t nmax = 2 * (node_ptr[i + 1] - node_ptr[i]);
    h_ptr[i + 1] = h_ptr[i] + nmax * nmax;
    if (nmax > KA_NLDS) need_aglob = true;
    largest = std::max(largest, nmax);
  }
  // LDS for the damped matrix is sized to the largest sub-problem (capped): smaller sub-problems
  // let several workgroups share a CU
  const int lds_n = (int)std::min<int64_t>(largest, KA_NLDS);
  const auto t0 = std::chrono::steady_clock::now();
  KaBuf<double> desc, kp_cand, vec, Hbuf, Abuf;
  KaBuf<int> var_of_node;
  KaBuf<uint8_t> used;
  KaBuf<int64_t> d_hptr;
  KaBuf<pxr_lm_summary> d_sum;
  const size_t nn = (size_t)view->n_nodes;
  if (int rc = desc.alloc(nn * 3 * arena->C)) return rc;
  if (int rc = kp_cand.alloc(nn * 2)) return rc;
  if (int rc = vec.alloc(nn * 2 * 10)) return rc;
  if (int rc = Hbuf.alloc((size_t)h_ptr[np])) return rc;
  if (int rc = Abuf.alloc(need_aglob ? (size_t)h_ptr[np] : 1)) return rc;
  if (int rc = var_of_node.alloc(nn)) return rc;
  if (int rc = used.alloc(nn)) return rc;
  if (int rc = d_hptr.alloc(np + 1)) return rc;
  if (int rc = d_sum.alloc(np)) return rc;
  PXR_HIP(hipMemsetAsync(used.p, 0, nn, st));
  PXR_HIP(hipMemcpyAsync(d_hptr.p, h_ptr.data(), sizeof(int64_t) * (np + 1), hipMemcpyHostToDevice, st));
  KaArgs a{};
  fill_args(ctx, arena, view, cfg, loss, a);
  a.bound = bound; a.opt = *options;
  a.desc = desc.p; a.kp_cand = kp_cand.p; a.var_of_node = var_of_node.p; a.used = used.p; a.vec = vec.p;
  a.prob_h_ptr = d_hptr.p; a.Hbuf = Hbuf.p; a.Abuf = Abuf.p; a.summaries = d_sum.p;
  a.lds_n = lds_n;
  const size_t shmem = sizeof(double) * (size_t)lds_n * lds_n;
#define KA_SOLVE_LAUNCH(ST, CC)                                                                              \
  do {                                                                                                       \
    PXR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ka_solve_kernel<ST, CC>),                      \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                    \
    hipLaunchKernelGGL((ka_solve_kernel<ST, CC>), dim3(np), dim3(256), shmem, st, a);                        \
  } while (0)
  if (arena->dtype == PXR_F16 && arena->C == 128) KA_SOLVE_LAUNCH(_Float16, 128);
  else if (arena->dtype == PXR_F16 && arena->C == 64) KA_SOLVE_LAUNCH(_Float16, 64);
  else if (arena->dtype == PXR_F32 && arena->C == 128) KA_SOLVE_LAUNCH(float, 128);
  else if (arena->dtype == PXR_F64 && arena->C == 128) KA_SOLVE_LAUNCH(double, 128);
  else return set_error(PXR_EUNSUPPORTED, "pxr_ka_solve: dtype/CHANNELS combination not supported (f16/f32/f64 x 128, f16 x 64)");
#undef KA_SOLVE_LAUNCH
  PXR_HIP(hipGetLastError());
  std::vector<pxr_lm_summary> sums(np);
  PXR_HIP(hipMemcpyAsync(sums.data(), d_sum.p, sizeof(pxr_lm_summary) * np, hipMemcpyDeviceToHost, st));
  PXR_HIP(hipStreamSynchronize(st));
  total->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  total->termination = PXR_TERM_CONVERGENCE;
  for (int i = 0; i < np; ++i) {   // AccumulateSummaries (util/src/statistics.h:131-160)
    total->initial_cost += sums[i].initial_cost; total->final_cost += sums[i].final_cost;
    total->iterations = std::max(total->iterations, sums[i].iterations);
    total->num_successful += sums[i].num_successful;
    total->num_point_unknowns += sums[i].num_camera_unknowns;
    if (sums[i].termination == PXR_TERM_FAILURE) total->termination = PXR_TERM_FAILURE;
    else if (sums[i].termination == PXR_TERM_NO_CONVERGENCE && total->termination != PXR_TERM_FAILURE)
      total->termination = PXR_TERM_NO_CONVERGENCE;
  }
  if (h_summaries) memcpy(h_summaries, sums.data(), sizeof(pxr_lm_summary) * np);
  return PXR_OK;
}

// pxr_ba_inner.hip -- Ceres-style inner iterations for featuremetric BA on gfx950.
//
// pixsfm enables `use_inner_iterations` for bundle adjustment by default
// (bundle_adjustment/main.py:43) with every variable point in inner-iteration group 0
// (bundle_adjustment/src/bundle_optimizer.h:350-355).  [upstream Ceres 2.1
// coordinate_descent_minimizer.cc, trust_region_minimizer.cc::DoInnerIterationsIfNeeded]: after
// the trust-region step, each point of the independent set is re-optimised on its own -- cameras
// fixed at the candidate -- by a nested TR-LM with Ceres' default options (<= 50 iterations,
// function / gradient / parameter tolerance 1e-6 / 1e-10 / 1e-8, initial radius 1e4, Jacobi
// scaling, <= 5 consecutive invalid steps).
//
// Mappings.  fp16 / fp32 feature patches (C = 64, 128), tracks of <= 16 observations, the five hand-derived camera models:
// k_inner_gram_packed -- the nested LM on the Gram matrices of the observations' stencils, up to four points per wavefront in
// lockstep, four lanes per observation (described at the kernel; k_inner_gram is its one-point-per-wavefront predecessor,
// kept as an A/B knob).  Longer tracks and extended camera models: k_inner_packed -- the descriptor interpolated at every
// round, 16 observations per wavefront trip, four lanes each.  fp64 storage and cost maps (C = 1, 3): k_inner_points -- one
// point per wavefront (eight for cost maps), an observation per row of C / 8 lanes, the whole nested LM in registers.  All
// return the cost at the (unrefined) candidate, so the outer loop needs no separate evaluation for Ceres' inner-iteration
// bookkeeping.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>
#include <vector>
#include <algorithm>

#include "pxr_device.h"
#include "pxr_gram.h"
#include "pxr_interp.h"
#include "pxr_internal.h"

namespace pxr {

struct InnerArgs {
  pxr_ba_view v;               // candidate parameters; d_xyz is updated in place
  const void* arena; const int32_t* corners; const double* scales; int H, W;
  int l2_normalize, check_bounds;
  double up;                   // upsampling_factor_ of the patches (cost maps only; feature patches: 1)
  pxr_loss loss;
  const int64_t* pt_ptr; const int64_t* pt_obs; const int* pt_var;
  double* xyz_out;             // == v.d_xyz (mutable alias)
  double* cost_before;         // += sum of 0.5 rho at the unrefined candidate
  double* cost_pt;             // deterministic mode (else NULL): [n_points] the per-point costs instead, summed in index order by the caller
  double* gram_G;              // the solve's Gram-matrix cache (pxr_ba_gram.hip; NULL: none): [n_obs][176] and the cells they were
  int2* gram_cell;             // built for -- k_inner_gram copies a matrix instead of building it where the cell matches, and
  int gram_warm;               // writes back what it builds; gram_warm = 0: nothing cached yet (the first call of a solve)
};

// sum over the rows (LPO lanes each) of a point's lanes; every lane of a row holds the row's value
template <int LPO>
__device__ __forceinline__ double rows_sum(double v) {
  if (LPO == 1) return row8_sum(v);   // cost maps: 8 lanes per point, one observation per lane
  if (LPO == 8) v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// points per wavefront / observations per point staged in LDS
// ONE wavefront per workgroup.  The nested LMs of different points take different numbers of evaluations
// (3-10), and a workgroup keeps its slot until its slowest wavefront is done: with four points per workgroup the two
// wavefront slots of a SIMD were filled 72 % of the time (round 3 counters: SQ_WAVE_CYCLES against SQ_BUSY_CYCLES).
template <int C> struct InnerShape { static constexpr int PPW = C >= 64 ? 1 : 8, MAXO = C >= 64 ? 16 : 8, WPB = 1; };

template <typename ST, int C, bool FS>   // FS: InterpolationConfig.use_float_simd
__device__ __forceinline__ void inner_points_body(const InnerArgs& a, double (*sh_obs)[InnerShape<C>::MAXO][26]) {
  static_assert(C == 128 || C == 64 || C <= 4, "one observation per row of C / 8 lanes, or per lane (cost maps)");
  // features: one point per wavefront, an observation per row of C / 8 lanes.  Cost maps (C = 1, 3;
  // costmap_bundle_optimizer.h:9-14): the whole texel in one lane, so a point takes 8 lanes (one DPP half-row: tracks
  // of up to 8 observations in one pass) and a wavefront runs 8 points, each group its own nested LM
  constexpr int PPW = InnerShape<C>::PPW, INNER_MAXO = InnerShape<C>::MAXO, GL = 64 / PPW, WPB = InnerShape<C>::WPB;
  constexpr int LPO = C >= 64 ? C / 8 : 1, ROWS = GL / LPO, CH = C >= 64 ? 8 : C;
  const int lane = (threadIdx.x & 63) % GL, row = lane / LPO, sub = lane % LPO;
  const int slot = (threadIdx.x >> 6) * PPW + (threadIdx.x & 63) / GL;   // this point's staging slot in LDS
  const int64_t p = (int64_t)blockIdx.x * WPB * PPW + slot;
  if (p >= a.v.n_points) return;
  const int64_t o0 = a.pt_ptr[p];
  const int n = (int)(a.pt_ptr[p + 1] - o0);
  if (n == 0) return;
  const bool variable = a.pt_var[p] != 0;
  double X[3] = {a.v.d_xyz[3 * p], a.v.d_xyz[3 * p + 1], a.v.d_xyz[3 * p + 2]};
  const ST* arena = reinterpret_cast<const ST*>(a.arena);
  const size_t patch_elems = (size_t)a.H * a.W * C;
  double ref[CH];   // d_refs == NULL: no reference is subtracted (cost maps)
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) ref[ch] = a.v.d_refs ? a.v.d_refs[(size_t)p * C + sub * CH + ch] : 0.0;

  auto rsum = [](double v) { return LPO == 16 ? row16_sum(v) : (LPO == 8 ? row8_sum(v) : v); };
  // stage the observations' camera / patch data (lane 0 of each row, one observation each)
  for (int oi = row; oi < n && oi < INNER_MAXO; oi += ROWS) {
    if (sub == 0) {
      double* ob = sh_obs[slot][oi];
      const int64_t i = a.pt_obs[o0 + oi];
      const int img = a.v.d_obs_image[i], cam = a.v.d_image_camera[img];
      const int64_t pi = a.v.d_obs_patch[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) ob[j] = a.v.d_qvec[4 * (size_t)img + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) ob[4 + j] = a.v.d_tvec[3 * (size_t)img + j];
#pragma unroll
      for (int j = 0; j < PXR_KPAD; ++j) ob[7 + j] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + j];
      ob[19] = a.scales[2 * pi]; ob[20] = a.scales[2 * pi + 1];
      ob[21] = (double)a.corners[2 * pi]; ob[22] = (double)a.corners[2 * pi + 1];
      ob[23] = (double)a.v.d_cam_model[cam]; ob[24] = (double)pi;
    }
  }
  __threadfence_block();                 // lane-0 writes -> visible to the other lanes of this wavefront
  __builtin_amdgcn_wave_barrier();

  // cost (+ normal equations H (6: xx xy xz yy yz zz), g (3)) of this point at Xc
  auto eval = [&](const double* Xc, bool with_jac, double* Hn, double* gn) -> double {
    double cost = 0.0, acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int chunk = 0; chunk < n; chunk += ROWS) {
      const int oi = chunk + row;
      const bool valid = oi < n;
      const int oc = valid ? oi : n - 1;
      // camera / patch data of the observation: staged once per point in LDS (the nested LM evaluates
      // the same observations ~10 times; the obs -> image -> camera -> parameters chain of dependent
      // global loads was most of every evaluation's latency); tracks longer than INNER_MAXO observations
      // read the tail from global memory
      double q[4], t[3], k[PXR_KPAD], sx, sy, cx, cy;
      int model;
      int64_t pi;
      if (oc < INNER_MAXO) {
        const double* ob = sh_obs[slot][oc];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = ob[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) t[j] = ob[4 + j];
#pragma unroll
        for (int j = 0; j < PXR_KPAD; ++j) k[j] = ob[7 + j];
        sx = ob[19]; sy = ob[20]; cx = ob[21]; cy = ob[22];
        model = (int)ob[23]; pi = (int64_t)ob[24];
      } else {
        const int64_t i = a.pt_obs[o0 + oc];
        const int img = a.v.d_obs_image[i], cam = a.v.d_image_camera[img];
        pi = a.v.d_obs_patch[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = a.v.d_qvec[4 * (size_t)img + j];
#pragma unroll
        for (int j = 0; j < 3; ++j) t[j] = a.v.d_tvec[3 * (size_t)img + j];
#pragma unroll
        for (int j = 0; j < PXR_KPAD; ++j) k[j] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + j];
        model = a.v.d_cam_model[cam];
        sx = a.scales[2 * pi]; sy = a.scales[2 * pi + 1];
        cx = (double)a.corners[2 * pi]; cy = (double)a.corners[2 * pi + 1];
      }
      double x, y, A[2][3], Pq[2][4], PX[2][3], Pk[2][PXR_KPAD];
      world_to_pixel_jac(model, k, q, t, Xc, x, y, A, Pq, PX, Pk);
      // FeaturePatch::ToPixelCoordinates (featurepatch.h:250-255); cost maps may carry an upsampling factor
      double u = x * sx - 0.5 - cx, v = y * sy - 0.5 - cy;
      if constexpr (C < 64) { u *= a.up; v *= a.up; sx *= a.up; sy *= a.up; }   // sx, sy: chain-rule factors from here on
      double f[CH], fr[CH], fc[CH];
      if constexpr (C >= 64)
        interp8<ST, LPO, true, FS>(arena + (size_t)pi * patch_elems, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f, fr, fc);
      else
        interp_small<ST, C>(arena + (size_t)pi * patch_elems, a.H, a.W, u, v, a.l2_normalize != 0, f, fr, fc);
      double s = 0, gcc = 0, gcr = 0, grr = 0, bc = 0, br = 0;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        const double r = f[ch] - ref[ch];
        s = fma(r, r, s);
        gcc = fma(fc[ch], fc[ch], gcc); gcr = fma(fc[ch], fr[ch], gcr); grr = fma(fr[ch], fr[ch], grr);
        bc = fma(fc[ch], r, bc); br = fma(fr[ch], r, br);
      }
      s = rsum(s);
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, s, rho);
      if (valid) cost += 0.5 * rho[0];
      // check_bounds: without a reference descriptor (cost maps) a projection outside its patch fails the evaluation
      // (feature_reference.h:128-130) -> non-finite cost -> the step is rejected
      if (valid && a.check_bounds && !a.v.d_refs && !(u > 0.0 && u < (double)a.W && v > 0.0 && v < (double)a.H)) cost = __builtin_nan("");
      if (with_jac) {
        gcc = rsum(gcc) * sx * sx; gcr = rsum(gcr) * sx * sy; grr = rsum(grr) * sy * sy;
        bc = rsum(bc) * sx; br = rsum(br) * sy;
        double kappa = 0.0;
        if (s != 0.0 && rho[2] > 0.0) {
          const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
          const double alpha = 1.0 - sqrt(D);
          kappa = (2.0 * alpha - alpha * alpha) / s;
        }
        const double w8 = valid ? rho[1] : 0.0;
        const double m00 = w8 * (gcc - kappa * bc * bc), m01 = w8 * (gcr - kappa * bc * br), m11 = w8 * (grr - kappa * br * br);
        const double b0 = w8 * bc, b1 = w8 * br;
        double me0[3], me1[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { me0[j] = m00 * PX[0][j] + m01 * PX[1][j]; me1[j] = m01 * PX[0][j] + m11 * PX[1][j]; }
        acc[0] += PX[0][0] * me0[0] + PX[1][0] * me1[0];
        acc[1] += PX[0][0] * me0[1] + PX[1][0] * me1[1];
        acc[2] += PX[0][0] * me0[2] + PX[1][0] * me1[2];
        acc[3] += PX[0][1] * me0[1] + PX[1][1] * me1[1];
        acc[4] += PX[0][1] * me0[2] + PX[1][1] * me1[2];
        acc[5] += PX[0][2] * me0[2] + PX[1][2] * me1[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[6 + j] += PX[0][j] * b0 + PX[1][j] * b1;
      }
    }
    cost = rows_sum<LPO>(cost);   // every lane of a row holds the row's cost -> sum of the 4 rows
    if (with_jac) {
#pragma unroll
      for (int j = 0; j < 6; ++j) Hn[j] = rows_sum<LPO>(acc[j]);
#pragma unroll
      for (int j = 0; j < 3; ++j) gn[j] = rows_sum<LPO>(acc[6 + j]);
    }
    return cost;
  };

  double H[6], g[3];
  double cost = eval(X, variable, H, g);
  if (lane == 0) { if (a.cost_pt) a.cost_pt[p] = cost; else atomicAdd(a.cost_before, cost); }
  if (!variable) return;
  // nested TR-LM (same loop as pxr_ba_solve, Ceres default options)
  double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  if (gmax <= 1e-10) return;
  const double sc[3] = {1.0 / (1.0 + sqrt(H[0])), 1.0 / (1.0 + sqrt(H[3])), 1.0 / (1.0 + sqrt(H[5]))};
  auto scale_sys = [&]() {
    H[0] *= sc[0] * sc[0]; H[1] *= sc[0] * sc[1]; H[2] *= sc[0] * sc[2];
    H[3] *= sc[1] * sc[1]; H[4] *= sc[1] * sc[2]; H[5] *= sc[2] * sc[2];
    g[0] *= sc[0]; g[1] *= sc[1]; g[2] *= sc[2];
  };
  scale_sys();
  double radius = 1e4, decrease_factor = 2.0, diag[3] = {0, 0, 0};
  int invalid = 0;
  bool reuse_diag = false;
  for (int it = 0; it < 50; ++it) {
    if (radius < 1e-32) break;
    if (!reuse_diag) {
      diag[0] = fmin(fmax(H[0], 1e-6), 1e32); diag[1] = fmin(fmax(H[3], 1e-6), 1e32); diag[2] = fmin(fmax(H[5], 1e-6), 1e32);
    }
    // Cholesky of the damped 3x3
    const double a00 = H[0] + diag[0] / radius, a01 = H[1], a02 = H[2], a11 = H[3] + diag[1] / radius, a12 = H[4],
                 a22 = H[5] + diag[2] / radius;
    bool ok = a00 > 0.0;
    const double l00 = sqrt(a00), l10 = a01 / l00, l20 = a02 / l00;
    const double d1 = a11 - l10 * l10;
    ok = ok && d1 > 0.0;
    const double l11 = sqrt(d1), l21 = (a12 - l20 * l10) / l11;
    const double d2 = a22 - l20 * l20 - l21 * l21;
    ok = ok && d2 > 0.0;
    const double l22 = sqrt(d2);
    const double y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11, y2 = (-g[2] - l20 * y0 - l21 * y1) / l22;
    double st[3];
    st[2] = y2 / l22; st[1] = (y1 - l21 * st[2]) / l11; st[0] = (y0 - l10 * st[1] - l20 * st[2]) / l00;
    double mcc = 0.0;
    if (ok) {
      const double dg = st[0] * g[0] + st[1] * g[1] + st[2] * g[2];
      const double dHd = st[0] * (H[0] * st[0] + H[1] * st[1] + H[2] * st[2]) + st[1] * (H[1] * st[0] + H[3] * st[1] + H[4] * st[2]) +
                         st[2] * (H[2] * st[0] + H[4] * st[1] + H[5] * st[2]);
      mcc = -dg - 0.5 * dHd;
      if (!(mcc > 0.0) || !isfinite(st[0]) || !isfinite(st[1]) || !isfinite(st[2])) ok = false;
    }
    if (!ok) {
      if (++invalid >= 5) break;
      radius *= 0.5; reuse_diag = true;
      continue;
    }
    invalid = 0;
    const double Xc[3] = {X[0] + st[0] * sc[0], X[1] + st[1] * sc[1], X[2] + st[2] * sc[2]};
    double Hc[6], gc[3];
    const double cand = eval(Xc, true, Hc, gc);   // with Jacobians: an accepted step needs no second evaluation
    const double s2 = (Xc[0] - X[0]) * (Xc[0] - X[0]) + (Xc[1] - X[1]) * (Xc[1] - X[1]) + (Xc[2] - X[2]) * (Xc[2] - X[2]);
    const double x2 = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
    if (sqrt(s2) <= 1e-8 * (sqrt(x2) + 1e-8)) break;
    const double cost_change = cost - cand;
    if (fabs(cost_change) <= 1e-6 * cost) break;
    const double rel = cost_change / mcc;
    if (rel > 1e-3) {
      X[0] = Xc[0]; X[1] = Xc[1]; X[2] = Xc[2];
      cost = cand;
#pragma unroll
      for (int j = 0; j < 6; ++j) H[j] = Hc[j];
#pragma unroll
      for (int j = 0; j < 3; ++j) g[j] = gc[j];
      gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
      scale_sys();
      const double tmp = 2.0 * rel - 1.0;
      radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp));
      decrease_factor = 2.0; reuse_diag = false;
      if (gmax <= 1e-10) break;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diag = true;
    }
  }
  if (lane == 0) { a.xyz_out[3 * p] = X[0]; a.xyz_out[3 * p + 1] = X[1]; a.xyz_out[3 * p + 2] = X[2]; }
}

// fp64 storage and the cost maps' few channels (the fp16 / fp32 feature patches take k_inner_packed below)
template <typename ST, int C, bool FS>
__global__ __launch_bounds__(64 * InnerShape<C>::WPB) void k_inner_points(const InnerArgs a) {
  __shared__ double sh_obs[InnerShape<C>::WPB * InnerShape<C>::PPW][InnerShape<C>::MAXO][26];
  inner_points_body<ST, C, FS>(a, sh_obs);
}

// ---- feature patches (C = 64, 128; fp16 / fp32 storage): observations packed 16 to a wavefront ------------------------------
// The mapping above spends a whole wavefront trip on at most C / 8-lane rows of ONE point: a track of five observations
// takes two trips (the second with one row busy), and every trip repeats the projection on all lanes.  Round 3 counters put
// the kernel at ~70 % vector-ALU utilisation with 14 600 instructions per point, so the instruction count IS the time.
// Here an observation takes FOUR lanes (each lane C / 32 chunks of 8 channels, the four lanes reading 64 consecutive bytes
// of a texel), a wavefront trip takes 16 observations, and a wavefront owns `ppw` consecutive points (16 / mean track
// length, at most 4) whose observations are consecutive in the point-major list: three 5-observation tracks per trip.
// Lane j < ppw runs the nested LM of point j (all points of the wavefront in lockstep: one evaluation per round for every
// point still iterating), the slots send their 3 x 3 contributions to the owners through LDS.
//
// Because a lane now walks several channel chunks, the descriptor cannot be normalised before the residual is formed
// without keeping C / 4 interpolated values and gradients per lane.  The sums the normal equations need are instead taken
// over the RAW interpolated f, df/dc, df/dr (and the reference d) and the normalisation is applied to the sums:
//   N^2 = f.f   r = f / N - d   r.r = 1 - 2 (f.d) / N + d.d
//   Jc = (fc - f (f.fc) / N^2) / N   Jc.Jc = (fc.fc - (f.fc)^2 / N^2) / N^2   Jc.r = -(fc.d - (f.d)(f.fc) / N^2) / N
// -- the same quantities as PixelInterpolator's normalise-then-subtract (interpolation.h:648-666) up to rounding: r.r is a
// difference of O(1) terms, relative error ~ 2e-16 / |r|^2 (1e-14 at |r| = 0.1, 1e-10 at |r| = 0.003;
// tests/test_inner_sums_identity.py); it is clamped at 0.  The nested LM is Ceres' heuristic refinement of a candidate the
// outer loop re-evaluates with the exact-order kernel, and its own tolerances are 1e-6 relative.
constexpr int IP_MAXPTS = 4, IP_MAXQ = 32, IP_OBS = 30;

// Nested-LM state of one point.  It lives in LDS: the owner lane works on it for a few hundred cycles per round, and as
// registers it was what the compiler spilled to scratch around every evaluation (101 VGPRs; the reloads sat on the round's
// critical path).
struct InnerOwner {
  double X[3], Xc[3], cost, H[6], g[3], sc[3], radius, decrease_factor, diag[3], mcc, r2;
  int invalid, it, reuse_diag, live;
};

template <int C, int IP_SLOTS>
struct PackedLds {
  double ref[IP_MAXPTS][C];          // reference descriptors of the wavefront's points
  double obs[IP_MAXQ][IP_OBS];       // per observation: R (9, row-major) t (3) k (12) sx sy corner (2) model patch
  double res[IP_SLOTS][10];          // per slot of the current trip: cost, H (6), g (3)
  InnerOwner own[IP_MAXPTS];
};

// 1 / sqrt(x) for finite x > 0 (v_rsq_f64 + one second-order correction); x <= 0 gives NaN / inf, which the callers test for
__device__ __forceinline__ double inner_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y0), y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
// 1 / x (v_rcp_f64 + two Newton steps, < 1 ulp for normal x): the IEEE division sequence is three times as long a chain
__device__ __forceinline__ double inner_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  return fma(fma(-x, y, 1.0), y, y);
}

__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  return v;
}

// rho(s) = {rho, rho', rho''} like loss_eval (pxr_device.h) with reciprocals for the divisions
__device__ __forceinline__ void inner_loss(int type, double a, double s, double rho[3]) {
  const double b = a * a;
  const double tiny = 2.2250738585072014e-308;
  switch (type) {
    case PXR_LOSS_CAUCHY: {
      const double c = inner_rcp(b), sum = 1.0 + s * c, inv = inner_rcp(sum);
      rho[0] = b * log(sum); rho[1] = fmax(tiny, inv); rho[2] = -c * (inv * inv);
      break;
    }
    case PXR_LOSS_HUBER:
      if (s > b) {
        const double ir = inner_rsqrt(s), r = s * ir;
        rho[0] = 2.0 * a * r - b; rho[1] = fmax(tiny, a * ir); rho[2] = -rho[1] * (0.5 * ir * ir);
      } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
      }
      break;
    case PXR_LOSS_SOFTL1: {
      const double c = inner_rcp(b), sum = 1.0 + s * c, it = inner_rsqrt(sum), tmp = sum * it;
      rho[0] = 2.0 * b * (tmp - 1.0); rho[1] = fmax(tiny, it);
      rho[2] = -(c * rho[1]) * (0.5 * it * it);
      break;
    }
    default:
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

// One round of a point's nested trust-region LM on its state in LDS (Ceres' loop as in inner_points_body): takes the
// evaluation (cand, Hc, gc) at the point's candidate, decides accept / reject / stop and forms the next candidate.
// first: the evaluation was at the unrefined point (its cost goes to *cost_before).
__device__ __forceinline__ void inner_owner_update(InnerOwner& S, const double cand, const double (&Hc)[6], const double (&gc)[3],
                                                   const bool first, const bool has_obs, bool& active, double* cost_before,
                                                   double* cost_slot /* deterministic mode: this point's slot, else NULL */) {
  if (first) {
    S.cost = cand;
    if (has_obs) { if (cost_slot) *cost_slot = cand; else atomicAdd(cost_before, cand); }
    if (active) {
      const double gmax = fmax(fabs(gc[0]), fmax(fabs(gc[1]), fabs(gc[2])));
      if (gmax <= 1e-10) active = false;
      const double s0 = 1.0 / (1.0 + sqrt(Hc[0])), s1 = 1.0 / (1.0 + sqrt(Hc[3])), s2 = 1.0 / (1.0 + sqrt(Hc[5]));
      S.sc[0] = s0; S.sc[1] = s1; S.sc[2] = s2;
      S.H[0] = Hc[0] * (s0 * s0); S.H[1] = Hc[1] * (s0 * s1); S.H[2] = Hc[2] * (s0 * s2);
      S.H[3] = Hc[3] * (s1 * s1); S.H[4] = Hc[4] * (s1 * s2); S.H[5] = Hc[5] * (s2 * s2);
      S.g[0] = gc[0] * s0; S.g[1] = gc[1] * s1; S.g[2] = gc[2] * s2;
    }
  } else if (active) {
    const double X0 = S.X[0], X1 = S.X[1], X2 = S.X[2], C0 = S.Xc[0], C1 = S.Xc[1], C2 = S.Xc[2];
    const double s2 = (C0 - X0) * (C0 - X0) + (C1 - X1) * (C1 - X1) + (C2 - X2) * (C2 - X2);
    const double x2 = X0 * X0 + X1 * X1 + X2 * X2;
    const double cost = S.cost, cost_change = cost - cand;
    if (sqrt(s2) <= 1e-8 * (sqrt(x2) + 1e-8)) active = false;
    else if (fabs(cost_change) <= 1e-6 * cost) active = false;
    else {
      const double rel = cost_change / S.mcc;
      if (rel > 1e-3) {
        S.X[0] = C0; S.X[1] = C1; S.X[2] = C2;
        S.cost = cand;
        const double s0 = S.sc[0], s1 = S.sc[1], s2c = S.sc[2];
        const double gmax = fmax(fabs(gc[0]), fmax(fabs(gc[1]), fabs(gc[2])));
        S.H[0] = Hc[0] * (s0 * s0); S.H[1] = Hc[1] * (s0 * s1); S.H[2] = Hc[2] * (s0 * s2c);
        S.H[3] = Hc[3] * (s1 * s1); S.H[4] = Hc[4] * (s1 * s2c); S.H[5] = Hc[5] * (s2c * s2c);
        S.g[0] = gc[0] * s0; S.g[1] = gc[1] * s1; S.g[2] = gc[2] * s2c;
        const double tmp = 2.0 * rel - 1.0;
        S.radius = fmin(1e16, S.radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp));
        S.decrease_factor = 2.0; S.reuse_diag = 0;
        if (gmax <= 1e-10) active = false;
      } else {
        const double df = S.decrease_factor;
        S.radius = S.radius / df; S.decrease_factor = df * 2.0; S.reuse_diag = 1;
      }
    }
  }
  // the next candidate (steps that fail without an evaluation are retried here, like the `continue` above)
  if (active) {
    const double H0 = S.H[0], H1 = S.H[1], H2 = S.H[2], H3 = S.H[3], H4 = S.H[4], H5 = S.H[5];
    const double g0 = S.g[0], g1 = S.g[1], g2 = S.g[2];
    double radius = S.radius, d0 = S.diag[0], d1v = S.diag[1], d2v = S.diag[2];
    int it = S.it, invalid = S.invalid;
    bool reuse_diag = S.reuse_diag != 0;
    while (true) {
      if (it >= 50 || radius < 1e-32) { active = false; break; }
      ++it;
      if (!reuse_diag) { d0 = fmin(fmax(H0, 1e-6), 1e32); d1v = fmin(fmax(H3, 1e-6), 1e32); d2v = fmin(fmax(H5, 1e-6), 1e32); }
      // Cholesky of the damped 3 x 3 with reciprocal pivots: three v_rsq_f64 and one division on the owners' chain instead
      // of three square roots and ten divisions
      const double ir = 1.0 / radius;
      const double a00 = H0 + d0 * ir, a11 = H3 + d1v * ir, a22 = H5 + d2v * ir;
      bool ok = a00 > 0.0;
      const double i00 = inner_rsqrt(a00), l10 = H1 * i00, l20 = H2 * i00;
      const double e1 = a11 - l10 * l10;
      ok = ok && e1 > 0.0;
      const double i11 = inner_rsqrt(e1), l21 = (H4 - l20 * l10) * i11;
      const double e2 = a22 - l20 * l20 - l21 * l21;
      ok = ok && e2 > 0.0;
      const double i22 = inner_rsqrt(e2);
      const double y0 = -g0 * i00, y1 = (-g1 - l10 * y0) * i11, y2 = (-g2 - l20 * y0 - l21 * y1) * i22;
      const double t2 = y2 * i22, t1 = (y1 - l21 * t2) * i11, t0 = (y0 - l10 * t1 - l20 * t2) * i00;
      double mcc = 0.0;
      if (ok) {
        const double dg = t0 * g0 + t1 * g1 + t2 * g2;
        const double dHd = t0 * (H0 * t0 + H1 * t1 + H2 * t2) + t1 * (H1 * t0 + H3 * t1 + H4 * t2) + t2 * (H2 * t0 + H4 * t1 + H5 * t2);
        mcc = -dg - 0.5 * dHd;
        if (!(mcc > 0.0) || !isfinite(t0) || !isfinite(t1) || !isfinite(t2)) ok = false;
      }
      if (!ok) {
        if (++invalid >= 5) { active = false; break; }
        radius *= 0.5; reuse_diag = true;
        continue;
      }
      invalid = 0;
      S.mcc = mcc;
      S.Xc[0] = S.X[0] + t0 * S.sc[0]; S.Xc[1] = S.X[1] + t1 * S.sc[1]; S.Xc[2] = S.X[2] + t2 * S.sc[2];
      break;
    }
    S.radius = radius; S.diag[0] = d0; S.diag[1] = d1v; S.diag[2] = d2v;
    S.it = it; S.invalid = invalid; S.reuse_diag = reuse_diag ? 1 : 0;
  }
  S.live = active ? 1 : 0;
}

template <typename ST, int C, bool FS, int LPO>
__device__ __forceinline__ void inner_packed_body(const InnerArgs& a, const int ppw, const int* __restrict__ pt_list,
                                                  PackedLds<C, 64 / LPO>& lds) {
  static_assert(C == 128 || C == 64, "LPO lanes per observation, C / (8 LPO) chunks of 8 channels per lane");
  constexpr int NCHUNK = C / (8 * LPO), IP_SLOTS = 64 / LPO;
  const int lane = threadIdx.x, sidx = lane / LPO, sub = lane % LPO;
#ifdef PXR_INNER_PROFILE   // tools/inner_phase_probe.sh: two wavefronts print how their shader-clock cycles split over the phases
  long long pf_t = __builtin_amdgcn_s_memtime(), pf_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int pf_rounds = 0;
#define PF_MARK(k) do { const long long n_ = __builtin_amdgcn_s_memtime(); pf_acc[k] += n_ - pf_t; pf_t = n_; } while (0)
#else
#define PF_MARK(k) do { } while (0)
#endif
  auto lanes_sum = [](double v) { return LPO == 4 ? quad_sum(v) : row8_sum(v); };
  // pt_list: the wavefront's (one) point comes from a list -- the host sends the points with long tracks here, ppw = 1
  const int64_t P0 = pt_list ? (int64_t)pt_list[blockIdx.x] : (int64_t)blockIdx.x * ppw;
  const int npts = (int)((a.v.n_points - P0) < (int64_t)ppw ? (a.v.n_points - P0) : (int64_t)ppw);
  // the observations of points P0 .. P0 + npts - 1 are entries [o0, o0 + L) of pt_obs; point j starts at st_j
  const int64_t o0 = a.pt_ptr[P0];
  const int st1 = (int)(a.pt_ptr[P0 + (1 < npts ? 1 : npts)] - o0), st2 = (int)(a.pt_ptr[P0 + (2 < npts ? 2 : npts)] - o0),
            st3 = (int)(a.pt_ptr[P0 + (3 < npts ? 3 : npts)] - o0);
  const int L = (int)(a.pt_ptr[P0 + npts] - o0);
  if (L == 0) return;
  const bool owner = lane < npts;
  const int64_t myp = P0 + (owner ? lane : 0);
  const int my_b = owner ? (int)(a.pt_ptr[myp] - o0) : 0, my_e = owner ? (int)(a.pt_ptr[myp + 1] - o0) : 0;
  const bool variable = owner && my_e > my_b && a.pt_var[myp] != 0;
  bool active = variable, first = true;
  const ST* arena = reinterpret_cast<const ST*>(a.arena);
  const size_t patch_elems = (size_t)a.H * a.W * C;
  const bool l2 = a.l2_normalize != 0;

  // ---- staging: references, observation records (rotation matrix of the unit quaternion instead of the quaternion) ----
  for (int j = 0; j < npts; ++j)
    for (int ch = lane; ch < C; ch += 64) lds.ref[j][ch] = a.v.d_refs ? a.v.d_refs[(size_t)(P0 + j) * C + ch] : 0.0;
  for (int q = sidx; q < L && q < IP_MAXQ; q += IP_SLOTS) {
    double* ob = lds.obs[q];
    const int64_t i = a.pt_obs[o0 + q];
    const int img = a.v.d_obs_image[i], cam = a.v.d_image_camera[img];
    const int64_t pi = a.v.d_obs_patch[i];
    if (sub == 0) {
      double R[9];
      quat_to_rotation(a.v.d_qvec + 4 * (size_t)img, R);
#pragma unroll
      for (int m = 0; m < 9; ++m) ob[m] = R[m];
#pragma unroll
      for (int m = 0; m < 3; ++m) ob[9 + m] = a.v.d_tvec[3 * (size_t)img + m];
    } else if (sub == 1) {
#pragma unroll
      for (int m = 0; m < 6; ++m) ob[12 + m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
    } else if (sub == 2) {
#pragma unroll
      for (int m = 6; m < PXR_KPAD; ++m) ob[12 + m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
    } else if (sub == 3) {
      ob[24] = a.scales[2 * pi]; ob[25] = a.scales[2 * pi + 1];
      ob[26] = (double)a.corners[2 * pi]; ob[27] = (double)a.corners[2 * pi + 1];
      ob[28] = (double)a.v.d_cam_model[cam]; ob[29] = (double)pi;
    }
  }
  if (owner) {
    InnerOwner& S = lds.own[lane];
#pragma unroll
    for (int m = 0; m < 3; ++m) { S.X[m] = a.v.d_xyz[3 * myp + m]; S.Xc[m] = S.X[m]; }
    S.live = 1; S.radius = 1e4; S.decrease_factor = 2.0; S.invalid = 0; S.it = 0; S.reuse_diag = 0; S.mcc = 0.0;
  }
  __syncthreads();
  for (int j = 0; j < npts; ++j) {   // d.d of every reference, all lanes
    double r2 = 0.0;
    for (int ch = lane; ch < C; ch += 64) r2 = fma(lds.ref[j][ch], lds.ref[j][ch], r2);
    r2 = rows_sum<16>(row16_sum(r2));
    if (lane == 0) lds.own[j].r2 = r2;
  }
  PF_MARK(0);

  // ---- the nested LMs of the wavefront's points, one evaluation (all points) per round ----
  while (true) {
    __syncthreads();
    // -- evaluation: cost, H (xx xy xz yy yz zz), g of every point at its candidate --
    // A point that has finished keeps its slots but they skip the evaluation: the wavefront's rounds are those of its slowest
    // point, the texel traffic (~4 KB per observation and round through an L2 it does not fit) only that of the points
    // still iterating.
    double cand = 0.0, Hc[6] = {0, 0, 0, 0, 0, 0}, gc[3] = {0, 0, 0};
    for (int base = 0; base < L; base += IP_SLOTS) {
      const int q = base + sidx;
      const int qc = q < L ? q : L - 1;
      const int j = (qc >= st1) + (qc >= st2) + (qc >= st3);
      if (q < L && lds.own[j].live) {
        double R[9], t[3], k[PXR_KPAD], sx, sy, cx, cy;
        int model;
        int64_t pi;
        if (q < IP_MAXQ) {
          const double* ob = lds.obs[q];
#pragma unroll
          for (int m = 0; m < 9; ++m) R[m] = ob[m];
#pragma unroll
          for (int m = 0; m < 3; ++m) t[m] = ob[9 + m];
#pragma unroll
          for (int m = 0; m < PXR_KPAD; ++m) k[m] = ob[12 + m];
          sx = ob[24]; sy = ob[25]; cx = ob[26]; cy = ob[27];
          model = (int)ob[28]; pi = (int64_t)ob[29];
        } else {   // long tracks: the tail of the list comes from global memory
          const int64_t i = a.pt_obs[o0 + q];
          const int img = a.v.d_obs_image[i], cam = a.v.d_image_camera[img];
          pi = a.v.d_obs_patch[i];
          quat_to_rotation(a.v.d_qvec + 4 * (size_t)img, R);
#pragma unroll
          for (int m = 0; m < 3; ++m) t[m] = a.v.d_tvec[3 * (size_t)img + m];
#pragma unroll
          for (int m = 0; m < PXR_KPAD; ++m) k[m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
          model = a.v.d_cam_model[cam];
          sx = a.scales[2 * pi]; sy = a.scales[2 * pi + 1];
          cx = (double)a.corners[2 * pi]; cy = (double)a.corners[2 * pi + 1];
        }
        // WorldToPixel (base/src/projection.h:60-75) and d(x,y)/dX = d(x,y)/d(u,v) d(u,v)/dp R
        const double X0 = lds.own[j].Xc[0], X1 = lds.own[j].Xc[1], X2 = lds.own[j].Xc[2];
        const double p0 = fma(R[0], X0, fma(R[1], X1, fma(R[2], X2, t[0])));
        const double p1 = fma(R[3], X0, fma(R[4], X1, fma(R[5], X2, t[1])));
        const double p2 = fma(R[6], X0, fma(R[7], X1, fma(R[8], X2, t[2])));
        const double iz = inner_rcp(p2), un = p0 * iz, vn = p1 * iz;
        double x, y, Juv[2][2];
        camera_model_jac<false>(model, k, un, vn, x, y, Juv, nullptr);
        double PX[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const double A0 = Juv[r][0] * iz, A1 = Juv[r][1] * iz, A2 = -(Juv[r][0] * un + Juv[r][1] * vn) * iz;
#pragma unroll
          for (int m = 0; m < 3; ++m) PX[r][m] = A0 * R[m] + A1 * R[3 + m] + A2 * R[6 + m];
        }
        const double u = x * sx - 0.5 - cx, v = y * sy - 0.5 - cy;   // FeaturePatch::ToPixelCoordinates (featurepatch.h:250-255)
        const ST* patch = arena + (size_t)pi * patch_elems;
        PF_MARK(1);
        // sums over this lane's channels: g = f (normalised mode) or f - d
        double Sgg = 0, Sgc = 0, Sgr = 0, Scc = 0, Scr = 0, Srr = 0, Sfd = 0, Scd = 0, Srd = 0;
#pragma unroll 1   // (unrolled, with the loads of chunk c + 1 issued between the two passes of chunk c: 54 more spilled registers, 4.02 -> 4.50 ms)
        for (int c = 0; c < NCHUNK; ++c) {
          const int chan0 = (c * LPO + sub) * 8;
          double f[8], fr[8], fc[8];
          interp8_raw<ST, true, FS>(patch, a.H, a.W, C, chan0, u, v, f, fr, fc);
          const double* rp = lds.ref[j] + chan0;
          if (l2) {
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
              const double d = rp[ch];
              Sgg = fma(f[ch], f[ch], Sgg); Sgc = fma(f[ch], fc[ch], Sgc); Sgr = fma(f[ch], fr[ch], Sgr);
              Scc = fma(fc[ch], fc[ch], Scc); Scr = fma(fc[ch], fr[ch], Scr); Srr = fma(fr[ch], fr[ch], Srr);
              Sfd = fma(f[ch], d, Sfd); Scd = fma(fc[ch], d, Scd); Srd = fma(fr[ch], d, Srd);
            }
          } else {
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
              const double r = f[ch] - rp[ch];
              Sgg = fma(r, r, Sgg); Sgc = fma(r, fc[ch], Sgc); Sgr = fma(r, fr[ch], Sgr);
              Scc = fma(fc[ch], fc[ch], Scc); Scr = fma(fc[ch], fr[ch], Scr); Srr = fma(fr[ch], fr[ch], Srr);
            }
          }
        }
        PF_MARK(2);
        Sgg = lanes_sum(Sgg); Sgc = lanes_sum(Sgc); Sgr = lanes_sum(Sgr);
        Scc = lanes_sum(Scc); Scr = lanes_sum(Scr); Srr = lanes_sum(Srr);
        double s, gcc, gcr, grr, bc, br;
        if (l2) {
          Sfd = lanes_sum(Sfd); Scd = lanes_sum(Scd); Srd = lanes_sum(Srd);
          const double ninv = inner_rsqrt(Sgg), n2inv = ninv * ninv;
          const double pc = Sgc * n2inv, pr = Sgr * n2inv;
          s = fmax(0.0, 1.0 - 2.0 * Sfd * ninv + lds.own[j].r2);
          gcc = (Scc - Sgc * pc) * n2inv; gcr = (Scr - Sgc * pr) * n2inv; grr = (Srr - Sgr * pr) * n2inv;
          bc = -(Scd - Sfd * pc) * ninv; br = -(Srd - Sfd * pr) * ninv;
        } else {
          s = Sgg; gcc = Scc; gcr = Scr; grr = Srr; bc = Sgc; br = Sgr;
        }
        double rho[3];
        inner_loss(a.loss.type, a.loss.a, s, rho);
        double cq = 0.5 * rho[0];
        // check_bounds: without a reference descriptor a projection outside its patch fails the evaluation
        // (feature_reference.h:128-130) -> non-finite cost -> the step is rejected
        if (a.check_bounds && !a.v.d_refs && !(u > 0.0 && u < (double)a.W && v > 0.0 && v < (double)a.H)) cq = __builtin_nan("");
        gcc *= sx * sx; gcr *= sx * sy; grr *= sy * sy; bc *= sx; br *= sy;
        double kappa = 0.0;   // Ceres' corrector (corrector.cc): alpha = 1 - sqrt(1 + 2 s rho'' / rho')
        if (s != 0.0 && rho[2] > 0.0) {
          const double D = 1.0 + 2.0 * s * rho[2] * inner_rcp(rho[1]);
          const double alpha = 1.0 - sqrt(D);
          kappa = (2.0 * alpha - alpha * alpha) * inner_rcp(s);
        }
        const double w8 = rho[1];
        const double m00 = w8 * (gcc - kappa * bc * bc), m01 = w8 * (gcr - kappa * bc * br), m11 = w8 * (grr - kappa * br * br);
        const double b0 = w8 * bc, b1 = w8 * br;
        double me0[3], me1[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { me0[m] = m00 * PX[0][m] + m01 * PX[1][m]; me1[m] = m01 * PX[0][m] + m11 * PX[1][m]; }
        if (sub == 0) {
          double* rs = lds.res[sidx];
          rs[0] = cq;
          rs[1] = PX[0][0] * me0[0] + PX[1][0] * me1[0];
          rs[2] = PX[0][0] * me0[1] + PX[1][0] * me1[1];
          rs[3] = PX[0][0] * me0[2] + PX[1][0] * me1[2];
          rs[4] = PX[0][1] * me0[1] + PX[1][1] * me1[1];
          rs[5] = PX[0][1] * me0[2] + PX[1][1] * me1[2];
          rs[6] = PX[0][2] * me0[2] + PX[1][2] * me1[2];
#pragma unroll
          for (int m = 0; m < 3; ++m) rs[7 + m] = PX[0][m] * b0 + PX[1][m] * b1;
        }
      }
      PF_MARK(3);
      __syncthreads();
      {   // owners: the slots of this trip that belong to their point, in track order
        const int lo = (my_b > base ? my_b : base) - base, hi = (my_e < base + IP_SLOTS ? my_e : base + IP_SLOTS) - base;
        for (int sl = lo; sl < hi; ++sl) {
          const double* rs = lds.res[sl];
          cand += rs[0];
#pragma unroll
          for (int m = 0; m < 6; ++m) Hc[m] += rs[1 + m];
#pragma unroll
          for (int m = 0; m < 3; ++m) gc[m] += rs[7 + m];
        }
      }
      if (base + IP_SLOTS < L) __syncthreads();   // the next trip overwrites res
    }
    PF_MARK(4);
    // -- owners: Ceres' trust-region bookkeeping (same loop as k_inner_points above) on the state in LDS --
    if (owner) inner_owner_update(lds.own[lane], cand, Hc, gc, first, my_e > my_b, active, a.cost_before, a.cost_pt ? a.cost_pt + myp : nullptr);
    first = false;
    PF_MARK(5);
#ifdef PXR_INNER_PROFILE
    ++pf_rounds;
#endif
    if (__ballot(active) == 0) break;
  }
#ifdef PXR_INNER_PROFILE
  if ((blockIdx.x == 1000 || blockIdx.x == 30000) && lane == 0)
    printf("[inner profile, wavefront %d, shader cycles] rounds %d  staging %lld  projection %lld  chunks %lld  tail %lld  owners' sums %lld  LM %lld\n",
           (int)blockIdx.x, pf_rounds, pf_acc[0], pf_acc[1], pf_acc[2], pf_acc[3], pf_acc[4], pf_acc[5]);
#endif
  if (variable) {
    const InnerOwner& S = lds.own[lane];
    a.xyz_out[3 * myp] = S.X[0]; a.xyz_out[3 * myp + 1] = S.X[1]; a.xyz_out[3 * myp + 2] = S.X[2];
  }
}

template <typename ST, int C, bool FS, int LPO>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_inner_packed(const InnerArgs a, const int ppw, const int* __restrict__ pt_list) {
  __shared__ PackedLds<C, 64 / LPO> lds;
  inner_packed_body<ST, C, FS, LPO>(a, ppw, pt_list, lds);
}

// ---- feature patches, third mapping: the nested LM on the GRAM MATRIX of each observation's stencil -------------------------
// k_inner_packed interpolates the 128-channel descriptor at every round of the nested LM: ~4 KB of texels per observation
// and round (13.7 GB fetched per call at configs[2], L2 hit rate 20-26 %: profiles/r3_inner_pmc_packed.json) and ~7 500
// vector instructions per point.  But a round never needs the descriptor itself -- only nine channel sums
//   f.f  f.fc  f.fr  fc.fc  fc.fr  fr.fr  f.d  fc.d  fr.d        (f, fc, fr: raw interpolated value / column / row derivative)
// and bicubic interpolation is linear in the 16 texels of the stencil:  f = sum_t w_t T_t,  w = wv (x) wu  (Catmull-Rom
// weights of the fractional position), fc: wv (x) wu',  fr: wv' (x) wu.  So with the stencil's Gram matrix G = T T^t (16 x 16)
// and D = T d (16) every sum is a quadratic / linear form in the weights:  f.f = w^t G w,  f.fc = w^t G wc, ...,  f.d = w.D.
// G and D are built ONCE per observation -- sixteen texels x C channels cross HBM once, the contraction over the channels is
// 32 v_mfma_f64_16x16x4 (A = B = one texel value per lane, fp64 accumulators: fp16 products are exact, the sums carry full
// double precision) -- and stay in LDS (2 KB per observation) while the point iterates.  One wavefront = one point; an
// observation takes 8 lanes, each with two rows of its G: a round is 88 fused multiply-adds and nine 8-lane reductions per
// lane instead of 2 048 texel loads' worth of splines.  The nested LM moves a point by a fraction of a texel, so the 4 x 4
// cell rarely changes between rounds; when it does (floor(u), floor(v) differ from the cached cell) that observation's Gram
// matrix is rebuilt.
//
// Arithmetic: the reference interpolates with an fp32 horizontal pass (cubic_hermite_spline_simd.h); here the whole bicubic
// is exact-in-fp64 algebra on the Gram matrix, so a round's cost differs from the exact-order kernels' by ~1e-7 relative
// (the fp32 pass's own rounding) -- inside the nested LM's 1e-6 tolerances, and like k_inner_packed's channel-sum
// normalisation it only steers Ceres' heuristic refinement: the outer loop re-evaluates the refined candidate with the
// exact-order kernel.  Points with more than IG_MAXO observations take k_inner_packed (the host splits the points).
#ifndef PXR_GRAM_WAVES
#define PXR_GRAM_WAVES 2     // wavefronts per SIMD the kernel is compiled for (tools/inner_gram_probe.sh builds the others)
#endif
constexpr int IG_MAXO = 16;   // observations per point whose Gram matrices fit the wavefront's LDS (two trips of 8)
constexpr int IG_OBS = 32;    // doubles of an observation record: IP_OBS + the cell its Gram matrix was built for
#ifndef PXR_GRAM_GSTRIDE
#define PXR_GRAM_GSTRIDE 160      // doubles between two observations' Gram matrices in LDS (>= 160; padding rotates the banks)
#endif
constexpr int IG_GSTRIDE = PXR_GRAM_GSTRIDE;
static_assert(IG_GSTRIDE >= IG_GDOUBLES, "ten 4 x 4 blocks per observation");   // (164 / 168 / 176: same 3.45 ms -- the 30 % LDS bank conflicts of the counters are not what bounds the kernel)

// one entry of the host's list of points for k_inner_gram
struct GramPoint { int p, len; int64_t o0; };
// per observation of a listed point: what the staging needs without walking obs -> image -> camera
struct GramSlot { int img, cam; int64_t patch; int64_t obs; };
// The kernel's table: entry e = {GramPoint, GramSlot[maxo]} at e * gram_entry_bytes(maxo) -- addressed by the workgroup index
// alone, so the list entry and its slots arrive in ONE round trip; the slots beyond the track's length repeat its last
// observation (a lane may read its slot before it knows the length).
__host__ __device__ inline size_t gram_entry_bytes(int maxo) { return sizeof(GramPoint) + sizeof(GramSlot) * (size_t)maxo; }
__global__ __launch_bounds__(256) void k_gram_table(int64_t n_list, int maxo, const GramPoint* __restrict__ pt_list, const int64_t* __restrict__ pt_obs,
                                                    const int32_t* __restrict__ obs_image, const int64_t* __restrict__ obs_patch,
                                                    const int32_t* __restrict__ image_camera, char* __restrict__ table) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = t / (maxo + 1);
  const int q = (int)(t % (maxo + 1)) - 1;                 // -1: the header
  if (e >= n_list) return;
  const GramPoint gp = pt_list[e];
  char* ent = table + (size_t)e * gram_entry_bytes(maxo);
  if (q < 0) { *reinterpret_cast<GramPoint*>(ent) = gp; return; }
  const int64_t i = pt_obs[gp.o0 + min(q, gp.len - 1)];
  const int img = obs_image[i];
  reinterpret_cast<GramSlot*>(ent + sizeof(GramPoint))[q] = GramSlot{img, image_camera[img], obs_patch[i], i};
}

// dynamic LDS of k_inner_gram for points of at most `maxo` observations, in doubles
__host__ __device__ inline size_t gram_lds_doubles(int maxo, int C) {
  return (size_t)maxo * (IG_GSTRIDE + 16 + IG_OBS) + C + (sizeof(InnerOwner) + 7) / 8;
}

template <typename ST, int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PXR_GRAM_WAVES, PXR_GRAM_WAVES))) void k_inner_gram(const InnerArgs a, const char* __restrict__ table, const int maxo) {
  static_assert(C == 128 || C == 64, "feature patches");
  extern __shared__ __align__(16) double gsh[];
  double* const Gs = gsh;                                  // [maxo][160]
  double* const Ds = Gs + (size_t)maxo * IG_GSTRIDE;      // [maxo][16]
  double* const obs = Ds + (size_t)maxo * 16;              // [maxo][IG_OBS]: R (9) t (3) k (12) sx sy corner (2) model patch cell (2)
  double* const refd = obs + (size_t)maxo * IG_OBS;        // [C] reference descriptor
  InnerOwner& S = *reinterpret_cast<InnerOwner*>(refd + C);
  const int lane = threadIdx.x;
#ifdef PXR_INNER_PROFILE   // tools/inner_phase_probe.sh: two wavefronts print how their shader-clock cycles split over the phases
  long long gq_t = __builtin_amdgcn_s_memtime(), gq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int gq_rounds = 0, gq_builds = 0;
#define GQ_MARK(k) do { const long long n_ = __builtin_amdgcn_s_memtime(); gq_acc[k] += n_ - gq_t; gq_t = n_; } while (0)
#else
#define GQ_MARK(k) do { } while (0)
#endif
  const char* const ent = table + (size_t)blockIdx.x * gram_entry_bytes(maxo);
  const GramSlot* const slots = reinterpret_cast<const GramSlot*>(ent + sizeof(GramPoint));   // this point's
  const GramPoint gp = *reinterpret_cast<const GramPoint*>(ent);
  const int64_t p = gp.p;
  const int L = gp.len;                                    // 1 .. maxo (the host's lists)
  const bool variable = a.pt_var[p] != 0;
  bool active = variable, first = true;                    // (meaningful on lane 0, the owner)
  const ST* arena = reinterpret_cast<const ST*>(a.arena);
  const size_t patch_elems = (size_t)a.H * a.W * C;
  const bool l2 = a.l2_normalize != 0;

  // ---- staging: reference, observation records (rotation matrix of the unit quaternion), cached Gram matrices, owner state.
  //      Two dependent round trips -- the table entry (list entry + slots, addressed by the workgroup index), then what the
  //      slots address, everything of it in flight together (the copies of the cached matrices are requested before the
  //      parameters and stored after them: the loads return in order).
  const int sq = lane >> 2, part = lane & 3;
  const GramSlot sl = slots[min(sq, maxo - 1)];            // (slots beyond the track's length repeat its last observation)
  double refv[C / 64];
#pragma unroll
  for (int j = 0; j < C / 64; ++j) refv[j] = a.v.d_refs ? a.v.d_refs[(size_t)p * C + lane + 64 * j] : 0.0;
  // The solve keeps every observation's Gram matrix for the cell of its last evaluation / inner call (pxr_ba_gram.hip): copy
  // the 1 408 bytes instead of reading 4 KB of texels and running 32 MFMAs.  Speculative -- the first round rebuilds the few
  // whose projection at the candidate left that cell.  Eight observations per batch (unconditional loads at clamped slots:
  // scalars, not a scratch array).
  const bool warm = a.gram_G != nullptr && a.gram_warm != 0;
  const int hi = lane < 24 ? lane : 23;
  double2 va0, va1, va2, va3, va4, va5, va6, va7, vb0, vb1, vb2, vb3, vb4, vb5, vb6, vb7;
#define IG_FETCH(Q0, J, VA, VB)                                                                                             \
  {                                                                                                                         \
    const int64_t oi = __shfl((int)sl.obs, 4 * min(Q0 + J, L - 1));      /* (observation indices fit 31 bits) */              \
    const double2* g = reinterpret_cast<const double2*>(a.gram_G + (size_t)oi * (IG_GDOUBLES + 16));                         \
    VA = g[lane]; VB = g[64 + hi];                                                                                          \
  }
#define IG_PUT(Q0, J, VA, VB)                                                                                               \
  if (Q0 + J < L) {                                                                                                         \
    double2* Gd = reinterpret_cast<double2*>(Gs + (size_t)(Q0 + J) * IG_GSTRIDE);                                           \
    Gd[lane] = VA;                                              /* doubles 0 .. 127 of G */                                 \
    if (lane < 16) Gd[64 + lane] = VB;                          /* 128 .. 159 */                                            \
    else if (lane < 24) reinterpret_cast<double2*>(Ds + (size_t)(Q0 + J) * 16)[lane - 16] = VB;   /* D */                   \
  }
#define IG_FETCH8(Q0) IG_FETCH(Q0, 0, va0, vb0) IG_FETCH(Q0, 1, va1, vb1) IG_FETCH(Q0, 2, va2, vb2) IG_FETCH(Q0, 3, va3, vb3) \
                      IG_FETCH(Q0, 4, va4, vb4) IG_FETCH(Q0, 5, va5, vb5) IG_FETCH(Q0, 6, va6, vb6) IG_FETCH(Q0, 7, va7, vb7)
#define IG_PUT8(Q0) IG_PUT(Q0, 0, va0, vb0) IG_PUT(Q0, 1, va1, vb1) IG_PUT(Q0, 2, va2, vb2) IG_PUT(Q0, 3, va3, vb3) \
                    IG_PUT(Q0, 4, va4, vb4) IG_PUT(Q0, 5, va5, vb5) IG_PUT(Q0, 6, va6, vb6) IG_PUT(Q0, 7, va7, vb7)
  if (warm) { IG_FETCH8(0) }
  if (sq < L) {
    double* ob = obs + (size_t)sq * IG_OBS;
    const int img = sl.img, cam = sl.cam;
    const int64_t pi = sl.patch;
    if (part == 0) {
      double R[9];
      quat_to_rotation(a.v.d_qvec + 4 * (size_t)img, R);
#pragma unroll
      for (int m = 0; m < 9; ++m) ob[m] = R[m];
#pragma unroll
      for (int m = 0; m < 3; ++m) ob[9 + m] = a.v.d_tvec[3 * (size_t)img + m];
    } else if (part == 1) {
#pragma unroll
      for (int m = 0; m < 6; ++m) ob[12 + m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
    } else if (part == 2) {
#pragma unroll
      for (int m = 6; m < PXR_KPAD; ++m) ob[12 + m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
    } else {
      ob[24] = a.scales[2 * pi]; ob[25] = a.scales[2 * pi + 1];
      ob[26] = (double)a.corners[2 * pi]; ob[27] = (double)a.corners[2 * pi + 1];
      ob[28] = (double)a.v.d_cam_model[cam]; ob[29] = (double)pi;
      // the cell its Gram matrix was built for: none yet -- or the cached matrix's
      int2 cc = make_int2(-1000000, -1000000);
      if (warm) cc = a.gram_cell[sl.obs];
      ob[30] = (double)cc.x; ob[31] = (double)cc.y;
    }
  }
#pragma unroll
  for (int j = 0; j < C / 64; ++j) refd[lane + 64 * j] = refv[j];
  if (warm) {
    IG_PUT8(0)
#pragma unroll 1
    for (int q0 = 8; q0 < L; q0 += 8) { IG_FETCH8(q0) IG_PUT8(q0) }
  }
#undef IG_FETCH
#undef IG_PUT
#undef IG_FETCH8
#undef IG_PUT8
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < 3; ++m) { S.X[m] = a.v.d_xyz[3 * p + m]; S.Xc[m] = S.X[m]; }
    S.live = 1; S.radius = 1e4; S.decrease_factor = 2.0; S.invalid = 0; S.it = 0; S.reuse_diag = 0; S.mcc = 0.0;
  }
  __syncthreads();
  {   // d.d of the reference
    double r2 = 0.0;
    for (int ch = lane; ch < C; ch += 64) r2 = fma(refd[ch], refd[ch], r2);
    r2 = rows_sum<16>(row16_sum(r2));
    if (lane == 0) S.r2 = r2;
  }

  // 8 lanes per observation, lane `sub` works on rows 2 sub, 2 sub + 1 of its Gram matrix; trip k: observations 8k .. 8k + 7
  const int sidx = lane >> 3, sub = lane & 7, ri = sub >> 1, ci = 2 * (sub & 1);
  GQ_MARK(0);

  while (true) {
    __syncthreads();
    double rs[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};       // this lane's observations: cost, H (6), g (3) (lanes with sub == 0)
#pragma unroll 1
    for (int k = 0; 8 * k < L; ++k) {
      const int q = 8 * k + sidx;
      const bool act = q < L;
      double* ob = obs + (size_t)(act ? q : L - 1) * IG_OBS;
      // WorldToPixel (base/src/projection.h:60-75) and d(x,y)/dX = d(x,y)/d(u,v) d(u,v)/dp R, on every lane of the observation
      double R[9];
#pragma unroll
      for (int m = 0; m < 9; ++m) R[m] = ob[m];
      const double sx = ob[24], sy = ob[25], cx = ob[26], cy = ob[27];
      const int model = (int)ob[28];
      const double X0 = S.Xc[0], X1 = S.Xc[1], X2 = S.Xc[2];
      const double p0 = fma(R[0], X0, fma(R[1], X1, fma(R[2], X2, ob[9])));
      const double p1 = fma(R[3], X0, fma(R[4], X1, fma(R[5], X2, ob[10])));
      const double p2 = fma(R[6], X0, fma(R[7], X1, fma(R[8], X2, ob[11])));
      const double iz = inner_rcp(p2), un = p0 * iz, vn = p1 * iz;
      double x, y, Juv[2][2];
      camera_model_jac<false, false>(model, ob + 12, un, vn, x, y, Juv, nullptr);   // (no extended models here: the host's routing)
      double PX[2][3];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double A0 = Juv[r][0] * iz, A1 = Juv[r][1] * iz, A2 = -(Juv[r][0] * un + Juv[r][1] * vn) * iz;
#pragma unroll
        for (int m = 0; m < 3; ++m) PX[r][m] = A0 * R[m] + A1 * R[3 + m] + A2 * R[6 + m];
      }
      const double u = x * sx - 0.5 - cx, v = y * sy - 0.5 - cy;   // FeaturePatch::ToPixelCoordinates (featurepatch.h:250-255)
      const double rf = floor(v), cf = floor(u);
      const int row = texel_index(rf, a.H), col = texel_index(cf, a.W);
      // -- (re)build the Gram matrices of this trip's observations whose cell moved: wave-uniform control flow, all 64 lanes;
      //    the texels of the next one are requested before the MFMA chain of the current one --
      const bool need = act && ((double)row != ob[30] || (double)col != ob[31]);
      unsigned long long todo = __ballot(need);
      GQ_MARK(1);
      if (todo != 0ull) {
        const int64_t pi = (int64_t)ob[29];
        const int pi_lo = (int)(pi & 0xffffffffll), pi_hi = (int)(pi >> 32);
        auto request = [&](GramTexels<ST, C>& tx, int src_lane) {
          const int row_t = __builtin_amdgcn_readlane(row, src_lane), col_t = __builtin_amdgcn_readlane(col, src_lane);
          const int64_t pi_t = ((int64_t)__builtin_amdgcn_readlane(pi_hi, src_lane) << 32) | (unsigned)__builtin_amdgcn_readlane(pi_lo, src_lane);
          tx.load(arena + (size_t)pi_t * patch_elems, a.H, a.W, row_t, col_t);
        };
        // a ring of three stencils in flight: a build's MFMA chain (~1 us) is shorter than the latency of its texels (~2 us under
        // load; with one stencil ahead every build waited for its loads: 6 000 cycles per build, profiles/r4_inner_gram_probe.txt)
        auto take = [&](GramTexels<ST, C>& tx) -> int {    // request the next observation's texels; -1: none left
          if (todo == 0ull) return -1;
          const int src = __ffsll((long long)todo) - 1;
          todo &= ~(0xffull << (src & ~7));
          request(tx, src);
          return src;
        };
        GramTexels<ST, C> t0, t1, t2;
        int s0 = take(t0), s1 = take(t1), s2 = take(t2);
        while (true) {
          if (s0 < 0) break;
          gram_contract<ST, C>(t0, refd, Gs + (size_t)(8 * k + (s0 >> 3)) * IG_GSTRIDE, Ds + (size_t)(8 * k + (s0 >> 3)) * 16);
          s0 = take(t0);
          if (s1 < 0) break;
          gram_contract<ST, C>(t1, refd, Gs + (size_t)(8 * k + (s1 >> 3)) * IG_GSTRIDE, Ds + (size_t)(8 * k + (s1 >> 3)) * 16);
          s1 = take(t1);
          if (s2 < 0) break;
          gram_contract<ST, C>(t2, refd, Gs + (size_t)(8 * k + (s2 >> 3)) * IG_GSTRIDE, Ds + (size_t)(8 * k + (s2 >> 3)) * 16);
          s2 = take(t2);
        }
        if (need && sub == 0) { ob[30] = (double)row; ob[31] = (double)col; }
        __syncthreads();                                 // the Gram matrices written by all lanes -> visible to their readers
        if (a.gram_G) {
          // ... and kept for the next call (and for the LM loop's evaluation, pxr_ba_gram.hip): an observation belongs to
          // one point, a point to one wavefront -- nobody else touches these 1 408 bytes
          unsigned long long wb = __ballot(need);
          while (wb != 0ull) {
            const int src = __ffsll((long long)wb) - 1;
            wb &= ~(0xffull << (src & ~7));
            const int slot = 8 * k + (src >> 3);
            const int64_t oi = slots[slot].obs;
            double2* g = reinterpret_cast<double2*>(a.gram_G + (size_t)oi * (IG_GDOUBLES + 16));
            const double2* Gd = reinterpret_cast<const double2*>(Gs + (size_t)slot * IG_GSTRIDE);
            g[lane] = Gd[lane];
            if (lane < 16) g[64 + lane] = Gd[64 + lane];
            else if (lane < 24) g[64 + lane] = reinterpret_cast<const double2*>(Ds + (size_t)slot * 16)[lane - 16];
            const int2 cl = make_int2(__builtin_amdgcn_readlane(row, src), __builtin_amdgcn_readlane(col, src));
            if (lane == 0) a.gram_cell[oi] = cl;
          }
        }
#ifdef PXR_INNER_PROFILE
        gq_builds += __popcll(__ballot(need)) / 8;
#endif
        GQ_MARK(2);
      }
      if (act) {
        double wu[4], dwu[4], wv[4], dwv[4];
        catmull_rom_weights(u - cf, wu, dwu);
        catmull_rom_weights(v - rf, wv, dwv);
        double ya[3], yb[3];                              // (G w, G wc, G wr) at rows 2 sub and 2 sub + 1
        gram_rows_times_weights(Gs + (size_t)q * IG_GSTRIDE, sub, wu, dwu, wv, dwv, ya, yb);
        const double2 d2 = *reinterpret_cast<const double2*>(Ds + (size_t)q * 16 + 2 * sub);
        const double wvo = pick4(wv, ri), dwvo = pick4(dwv, ri);
        const double wua = ci == 0 ? wu[0] : wu[2], wub = ci == 0 ? wu[1] : wu[3];
        const double dwua = ci == 0 ? dwu[0] : dwu[2], dwub = ci == 0 ? dwu[1] : dwu[3];
        const double oma = wvo * wua, omb = wvo * wub, omca = wvo * dwua, omcb = wvo * dwub, omra = dwvo * wua, omrb = dwvo * wub;
        const double Sgg = row8_sum(fma(oma, ya[0], omb * yb[0])), Sgc = row8_sum(fma(oma, ya[1], omb * yb[1]));
        const double Sgr = row8_sum(fma(oma, ya[2], omb * yb[2]));
        const double Scc = row8_sum(fma(omca, ya[1], omcb * yb[1])), Scr = row8_sum(fma(omca, ya[2], omcb * yb[2]));
        const double Srr = row8_sum(fma(omra, ya[2], omrb * yb[2]));
        const double Sfd = row8_sum(fma(oma, d2.x, omb * d2.y)), Scd = row8_sum(fma(omca, d2.x, omcb * d2.y));
        const double Srd = row8_sum(fma(omra, d2.x, omrb * d2.y));
        double s, gcc, gcr, grr, bc, br;
        if (l2) {
          const double ninv = inner_rsqrt(Sgg), n2inv = ninv * ninv;
          const double pc = Sgc * n2inv, pr = Sgr * n2inv;
          s = fmax(0.0, 1.0 - 2.0 * Sfd * ninv + S.r2);
          gcc = (Scc - Sgc * pc) * n2inv; gcr = (Scr - Sgc * pr) * n2inv; grr = (Srr - Sgr * pr) * n2inv;
          bc = -(Scd - Sfd * pc) * ninv; br = -(Srd - Sfd * pr) * ninv;
        } else {          // r = f - d
          s = fmax(0.0, Sgg - 2.0 * Sfd + S.r2);
          gcc = Scc; gcr = Scr; grr = Srr; bc = Sgc - Scd; br = Sgr - Srd;
        }
        double rho[3];
        inner_loss(a.loss.type, a.loss.a, s, rho);
        double cq = 0.5 * rho[0];
        if (a.check_bounds && !a.v.d_refs && !(u > 0.0 && u < (double)a.W && v > 0.0 && v < (double)a.H)) cq = __builtin_nan("");
        gcc *= sx * sx; gcr *= sx * sy; grr *= sy * sy; bc *= sx; br *= sy;
        double kappa = 0.0;   // Ceres' corrector (corrector.cc): alpha = 1 - sqrt(1 + 2 s rho'' / rho')
        if (s != 0.0 && rho[2] > 0.0) {
          const double D = 1.0 + 2.0 * s * rho[2] * inner_rcp(rho[1]);
          const double alpha = 1.0 - sqrt(D);
          kappa = (2.0 * alpha - alpha * alpha) * inner_rcp(s);
        }
        const double w8 = rho[1];
        const double m00 = w8 * (gcc - kappa * bc * bc), m01 = w8 * (gcr - kappa * bc * br), m11 = w8 * (grr - kappa * br * br);
        const double b0 = w8 * bc, b1 = w8 * br;
        double me0[3], me1[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) { me0[m] = m00 * PX[0][m] + m01 * PX[1][m]; me1[m] = m01 * PX[0][m] + m11 * PX[1][m]; }
        if (sub == 0) {
          rs[0] += cq;
          rs[1] += PX[0][0] * me0[0] + PX[1][0] * me1[0];
          rs[2] += PX[0][0] * me0[1] + PX[1][0] * me1[1];
          rs[3] += PX[0][0] * me0[2] + PX[1][0] * me1[2];
          rs[4] += PX[0][1] * me0[1] + PX[1][1] * me1[1];
          rs[5] += PX[0][1] * me0[2] + PX[1][1] * me1[2];
          rs[6] += PX[0][2] * me0[2] + PX[1][2] * me1[2];
#pragma unroll
          for (int m = 0; m < 3; ++m) rs[7 + m] += PX[0][m] * b0 + PX[1][m] * b1;
        }
      }
    }
    // -- the owner (lane 0): the sum over the observations (a fixed tree over lanes 0, 8, .., 56; no LDS round trip), then
    //    Ceres' trust-region bookkeeping --
    GQ_MARK(3);
#pragma unroll
    for (int m = 0; m < 10; ++m) {
      rs[m] += __shfl_xor(rs[m], 8); rs[m] += __shfl_xor(rs[m], 16); rs[m] += __shfl_xor(rs[m], 32);
    }
    GQ_MARK(4);
    if (lane == 0) {
      const double Hc[6] = {rs[1], rs[2], rs[3], rs[4], rs[5], rs[6]}, gc[3] = {rs[7], rs[8], rs[9]};
      inner_owner_update(S, rs[0], Hc, gc, first, true, active, a.cost_before, a.cost_pt ? a.cost_pt + p : nullptr);
    }
    first = false;
    GQ_MARK(5);
#ifdef PXR_INNER_PROFILE
    ++gq_rounds;
#endif
    if (__builtin_amdgcn_readfirstlane((int)active) == 0) break;
  }
#ifdef PXR_INNER_PROFILE
  if ((blockIdx.x == 1000 || blockIdx.x == 150000) && lane == 0)
    printf("[inner gram profile, wavefront %d, 100 MHz ticks x 10 ns] rounds %d builds %d  staging %lld  projection %lld  gram builds %lld  evaluation %lld  reduce %lld  owner LM %lld\n",
           (int)blockIdx.x, gq_rounds, gq_builds, gq_acc[0], gq_acc[1], gq_acc[2], gq_acc[3], gq_acc[4], gq_acc[5]);
#endif
  if (variable && lane == 0) { a.xyz_out[3 * p] = S.X[0]; a.xyz_out[3 * p + 1] = S.X[1]; a.xyz_out[3 * p + 2] = S.X[2]; }
}

// ---- feature patches, fourth mapping: the Gram-matrix evaluation in the packed kernel's lockstep ------------------------------
// k_inner_gram is bound by its vector instruction count (profiles/r4_hot_kernels_pmc.json: ~3 900 wave instructions per point,
// the SIMDs 75 % busy issuing them): one point per wavefront repeats the projection, the weights, the robustifier and the
// owner's trust-region step for five observations on 40 lanes.  Here a wavefront takes up to four points whose observations
// fill at most sixteen slots (the host packs consecutive short tracks: three 5-observation tracks per wavefront), an
// observation takes FOUR lanes -- lane R works on block row R of its Gram matrix, rows 4 R .. 4 R + 3 -- and the points
// iterate in lockstep like in k_inner_packed: one round costs the same ~800 wave instructions whether it serves one point
// or four.  The nine sums are bilinear forms per 4 x 4 block:  B(a, b) = sum_ir sum_cc a[ir] G[4 R + ir][4 Cb + cc] b[cc]  with
// a, b the horizontal weights or their derivatives, scaled by the vertical weights of block row R and block column Cb; a block
// below the diagonal is the transposed stored block, i.e. the stored block's form with a and b swapped.
struct GramWave { int npts, pad; int p[IP_MAXPTS]; int st[IP_MAXPTS + 1]; int pad2; };   // points, first slot of each (st[npts] = slots used)
static_assert(sizeof(GramWave) == 48, "table layout");
constexpr int GW_SLOTS = 16;
__host__ __device__ inline size_t gram_wave_bytes() { return sizeof(GramWave) + sizeof(GramSlot) * GW_SLOTS; }
#ifndef PXR_GW_GPAD
#define PXR_GW_GPAD 4          // doubles of padding between two slots' matrices: 1 280-byte slots put the same block of all sixteen on the same banks (0 / 2 / 4 / 6: 1.71 / 1.59 / 1.58 / 1.59 ms per call)
#endif
struct GramWaveLds {
  double G[GW_SLOTS][IG_GDOUBLES + PXR_GW_GPAD];
  double D[GW_SLOTS][16];
  double obs[GW_SLOTS][IG_OBS];      // R (9) t (3) k (12) sx sy corner (2) model patch cell (2)
  double res[GW_SLOTS][10];
  double refbuf[128];                // reference descriptor of the observation being built
  InnerOwner own[IP_MAXPTS];
};

__global__ __launch_bounds__(256) void k_gram_wave_table(int64_t n_waves, const GramWave* __restrict__ waves, const int64_t* __restrict__ pt_ptr,
                                                         const int64_t* __restrict__ pt_obs, const int32_t* __restrict__ obs_image,
                                                         const int64_t* __restrict__ obs_patch, const int32_t* __restrict__ image_camera,
                                                         char* __restrict__ table) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = t / (GW_SLOTS + 1);
  const int q = (int)(t % (GW_SLOTS + 1)) - 1;             // -1: the header
  if (e >= n_waves) return;
  const GramWave w = waves[e];
  char* ent = table + (size_t)e * gram_wave_bytes();
  if (q < 0) { *reinterpret_cast<GramWave*>(ent) = w; return; }
  const int L = w.st[w.npts];
  const int qc = min(q, L - 1);                            // (slots beyond the last observation repeat it)
  int j = 0;
  while (j + 1 < w.npts && qc >= w.st[j + 1]) ++j;
  const int64_t i = pt_obs[pt_ptr[w.p[j]] + (qc - w.st[j])];
  const int img = obs_image[i];
  reinterpret_cast<GramSlot*>(ent + sizeof(GramWave))[q] = GramSlot{img, image_camera[img], obs_patch[i], i};
}

// Before a call of k_inner_gram_packed: which observations of its table project, at the candidate, into another cell than their
// cached Gram matrix was built for?  They are flagged (and their cell updated) for k_gram_build (pxr_ba_gram.hip), which
// rebuilds at the full rate of the matrix pipe -- four wavefronts per SIMD doing nothing else: 1.5 ms for every observation of
// configs[2] -- instead of inside the nested-LM kernel, where a wavefront's fifteen builds queue behind each other (the first
// call of a solve, every matrix cold: 4.5 ms).  The kernel then copies; it still builds what moves DURING its rounds.
__global__ __launch_bounds__(256) void k_gram_flag_slots(const InnerArgs a, const char* __restrict__ table, int64_t n_waves, int* __restrict__ dirty) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = t / GW_SLOTS;
  const int q = (int)(t % GW_SLOTS);
  if (e >= n_waves) return;
  const char* ent = table + (size_t)e * gram_wave_bytes();
  const GramWave* gw = reinterpret_cast<const GramWave*>(ent);
  const int npts = gw->npts, L = gw->st[npts];
  if (q >= L) return;
  int j = 0;
  while (j + 1 < npts && q >= gw->st[j + 1]) ++j;
  const int64_t p = gw->p[j];
  const GramSlot sl = reinterpret_cast<const GramSlot*>(ent + sizeof(GramWave))[q];
  double qv[4], tv[3], X[3], k[PXR_KPAD];
#pragma unroll
  for (int m = 0; m < 4; ++m) qv[m] = a.v.d_qvec[4 * (size_t)sl.img + m];
#pragma unroll
  for (int m = 0; m < 3; ++m) { tv[m] = a.v.d_tvec[3 * (size_t)sl.img + m]; X[m] = a.v.d_xyz[3 * (size_t)p + m]; }
#pragma unroll
  for (int m = 0; m < PXR_KPAD; ++m) k[m] = a.v.d_cam_params[(size_t)sl.cam * PXR_KPAD + m];
  // the projection of the nested LM's first round, operation for operation (rotation matrix of the unit quaternion, reciprocal
  // depth): the cell must be the one k_inner_gram_packed computes, or it would rebuild anyway
  double R[9];
  quat_to_rotation(qv, R);
  const double p0 = fma(R[0], X[0], fma(R[1], X[1], fma(R[2], X[2], tv[0])));
  const double p1 = fma(R[3], X[0], fma(R[4], X[1], fma(R[5], X[2], tv[1])));
  const double p2 = fma(R[6], X[0], fma(R[7], X[1], fma(R[8], X[2], tv[2])));
  const double iz = inner_rcp(p2), un = p0 * iz, vn = p1 * iz;
  double x, y, Juv[2][2];
  camera_model_jac<false, false>(a.v.d_cam_model[sl.cam], k, un, vn, x, y, Juv, nullptr);
  const double u = x * a.scales[2 * sl.patch] - 0.5 - (double)a.corners[2 * sl.patch];
  const double v = y * a.scales[2 * sl.patch + 1] - 0.5 - (double)a.corners[2 * sl.patch + 1];
  const int row = texel_index(floor(v), a.H), col = texel_index(floor(u), a.W);
  const int2 c = a.gram_cell[sl.obs];
  const bool miss = c.x != row || c.y != col;
  dirty[sl.obs] = miss ? 1 : 0;
  if (miss) a.gram_cell[sl.obs] = make_int2(row, col);
}

template <typename ST, int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_inner_gram_packed(const InnerArgs a, const char* __restrict__ table) {
  static_assert(C == 128 || C == 64, "feature patches");
  __shared__ __align__(16) GramWaveLds lds;
  const int lane = threadIdx.x, sidx = lane >> 2, sub = lane & 3;
  const char* const ent = table + (size_t)blockIdx.x * gram_wave_bytes();
  const GramWave* const gw = reinterpret_cast<const GramWave*>(ent);   // (indexed by lane: read where it lies, a private copy would live in scratch)
  const GramSlot* const slots = reinterpret_cast<const GramSlot*>(ent + sizeof(GramWave));
  const GramSlot sl = slots[sidx];                         // (slots beyond the last observation repeat it)
  const int npts = gw->npts, L = gw->st[npts];
  const int st1 = gw->st[1 < npts ? 1 : npts], st2 = gw->st[2 < npts ? 2 : npts], st3 = gw->st[3 < npts ? 3 : npts];
  const bool owner = lane < npts;
  const int64_t myp = gw->p[owner ? lane : 0];
  const int my_b = owner ? gw->st[lane] : 0, my_e = owner ? gw->st[lane + 1] : 0;
  const bool variable = owner && a.pt_var[myp] != 0;
  bool active = variable, first = true;
  const ST* arena = reinterpret_cast<const ST*>(a.arena);
  const size_t patch_elems = (size_t)a.H * a.W * C;
  const bool l2 = a.l2_normalize != 0;
  const bool warm = a.gram_G != nullptr && a.gram_warm != 0;
  const int q = sidx;
  const bool valid = q < L;
  const int jq = ((valid ? q : L - 1) >= st1) + ((valid ? q : L - 1) >= st2) + ((valid ? q : L - 1) >= st3);   // the point of this lane's slot

  // ---- staging: cached Gram matrices (requested first, stored last: the loads return in order), observation records, d.d of the
  //      references, owner state
  const int hi = lane < 24 ? lane : 23;
  double2 va0, va1, va2, va3, va4, va5, va6, va7, vb0, vb1, vb2, vb3, vb4, vb5, vb6, vb7;
#define GW_FETCH(Q0, J, VA, VB)                                                                                       \
  {                                                                                                                   \
    const int64_t oi = __shfl((int)sl.obs, 4 * min(Q0 + J, L - 1));      /* (observation indices fit 31 bits) */        \
    const double2* g = reinterpret_cast<const double2*>(a.gram_G + (size_t)oi * (IG_GDOUBLES + 16));                   \
    VA = g[lane]; VB = g[64 + hi];                                                                                    \
  }
#define GW_PUT(Q0, J, VA, VB)                                                                                         \
  if (Q0 + J < L) {                                                                                                   \
    double2* Gd = reinterpret_cast<double2*>(lds.G[Q0 + J]);                                                          \
    Gd[lane] = VA;                                                                                                    \
    if (lane < 16) Gd[64 + lane] = VB;                                                                                \
    else if (lane < 24) reinterpret_cast<double2*>(lds.D[Q0 + J])[lane - 16] = VB;                                    \
  }
#define GW_FETCH8(Q0) GW_FETCH(Q0, 0, va0, vb0) GW_FETCH(Q0, 1, va1, vb1) GW_FETCH(Q0, 2, va2, vb2) GW_FETCH(Q0, 3, va3, vb3) \
                      GW_FETCH(Q0, 4, va4, vb4) GW_FETCH(Q0, 5, va5, vb5) GW_FETCH(Q0, 6, va6, vb6) GW_FETCH(Q0, 7, va7, vb7)
#define GW_PUT8(Q0) GW_PUT(Q0, 0, va0, vb0) GW_PUT(Q0, 1, va1, vb1) GW_PUT(Q0, 2, va2, vb2) GW_PUT(Q0, 3, va3, vb3) \
                    GW_PUT(Q0, 4, va4, vb4) GW_PUT(Q0, 5, va5, vb5) GW_PUT(Q0, 6, va6, vb6) GW_PUT(Q0, 7, va7, vb7)
  if (warm) { GW_FETCH8(0) }
  if (valid) {
    double* ob = lds.obs[q];
    const int img = sl.img, cam = sl.cam;
    const int64_t pi = sl.patch;
    if (sub == 0) {
      double R[9];
      quat_to_rotation(a.v.d_qvec + 4 * (size_t)img, R);
#pragma unroll
      for (int m = 0; m < 9; ++m) ob[m] = R[m];
#pragma unroll
      for (int m = 0; m < 3; ++m) ob[9 + m] = a.v.d_tvec[3 * (size_t)img + m];
    } else if (sub == 1) {
#pragma unroll
      for (int m = 0; m < 6; ++m) ob[12 + m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
    } else if (sub == 2) {
#pragma unroll
      for (int m = 6; m < PXR_KPAD; ++m) ob[12 + m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
    } else {
      ob[24] = a.scales[2 * pi]; ob[25] = a.scales[2 * pi + 1];
      ob[26] = (double)a.corners[2 * pi]; ob[27] = (double)a.corners[2 * pi + 1];
      ob[28] = (double)a.v.d_cam_model[cam]; ob[29] = (double)pi;
      int2 cc = make_int2(-1000000, -1000000);            // the cell its Gram matrix was built for: none yet, or the cached matrix's
      if (warm) cc = a.gram_cell[sl.obs];
      ob[30] = (double)cc.x; ob[31] = (double)cc.y;
    }
  }
  for (int j = 0; j < npts; ++j) {                         // d.d of every reference, all lanes
    double r2 = 0.0;
    if (a.v.d_refs)
      for (int ch = lane; ch < C; ch += 64) { const double d = a.v.d_refs[(size_t)gw->p[j] * C + ch]; r2 = fma(d, d, r2); }
    r2 = rows_sum<16>(row16_sum(r2));
    if (lane == 0) lds.own[j].r2 = r2;
  }
  if (owner) {
    InnerOwner& S = lds.own[lane];
#pragma unroll
    for (int m = 0; m < 3; ++m) { S.X[m] = a.v.d_xyz[3 * myp + m]; S.Xc[m] = S.X[m]; }
    S.live = 1; S.radius = 1e4; S.decrease_factor = 2.0; S.invalid = 0; S.it = 0; S.reuse_diag = 0; S.mcc = 0.0;
  }
  if (warm) {
    GW_PUT8(0)
    if (L > 8) { GW_FETCH8(8) GW_PUT8(8) }
  }
#undef GW_FETCH
#undef GW_PUT
#undef GW_FETCH8
#undef GW_PUT8

  // ---- the nested LMs of the wavefront's points, one evaluation (all points) per round ----
  while (true) {
    __syncthreads();
    const bool act = valid && lds.own[jq].live != 0;       // a finished point's slots skip the round
    double* ob = lds.obs[valid ? q : L - 1];
    double PX[2][3], u = 0.0, v = 0.0, rf = 0.0, cf = 0.0, sx = 1.0, sy = 1.0;
    int row = 0, col = 0;
    if (act) {
      // WorldToPixel (base/src/projection.h:60-75) and d(x,y)/dX = d(x,y)/d(u,v) d(u,v)/dp R, on every lane of the observation
      double R[9];
#pragma unroll
      for (int m = 0; m < 9; ++m) R[m] = ob[m];
      sx = ob[24]; sy = ob[25];
      const double cx = ob[26], cy = ob[27];
      const int model = (int)ob[28];
      const double X0 = lds.own[jq].Xc[0], X1 = lds.own[jq].Xc[1], X2 = lds.own[jq].Xc[2];
      const double p0 = fma(R[0], X0, fma(R[1], X1, fma(R[2], X2, ob[9])));
      const double p1 = fma(R[3], X0, fma(R[4], X1, fma(R[5], X2, ob[10])));
      const double p2 = fma(R[6], X0, fma(R[7], X1, fma(R[8], X2, ob[11])));
      const double iz = inner_rcp(p2), un = p0 * iz, vn = p1 * iz;
      double x, y, Juv[2][2];
      camera_model_jac<false, false>(model, ob + 12, un, vn, x, y, Juv, nullptr);   // (no extended models here: the host's routing)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const double A0 = Juv[r][0] * iz, A1 = Juv[r][1] * iz, A2 = -(Juv[r][0] * un + Juv[r][1] * vn) * iz;
#pragma unroll
        for (int m = 0; m < 3; ++m) PX[r][m] = A0 * R[m] + A1 * R[3 + m] + A2 * R[6 + m];
      }
      u = x * sx - 0.5 - cx; v = y * sy - 0.5 - cy;        // FeaturePatch::ToPixelCoordinates (featurepatch.h:250-255)
      rf = floor(v); cf = floor(u);
      row = texel_index(rf, a.H); col = texel_index(cf, a.W);
    }
    // -- (re)build the Gram matrices of the observations whose cell moved: wave-uniform control flow, all 64 lanes --
    const bool need = act && ((double)row != ob[30] || (double)col != ob[31]);
    unsigned long long todo = __ballot(need);
    if (todo != 0ull) {
      const int64_t pi = (int64_t)ob[29];
      const int pi_lo = (int)(pi & 0xffffffffll), pi_hi = (int)(pi >> 32);
      const int pt_of_slot = gw->p[jq];
      const unsigned long long rebuilt = todo;
      // a ring of three stencils (and their references) in flight: a build's MFMA chain is shorter than the latency of its texels
      double2 rr0 = make_double2(0.0, 0.0), rr1 = rr0, rr2 = rr0;
      auto take = [&](GramTexels<ST, C>& tx, double2& rr) -> int {   // request the next observation's texels and reference; -1: none left
        if (todo == 0ull) return -1;
        const int src = __ffsll((long long)todo) - 1;
        todo &= ~(0xfull << (src & ~3));
        const int row_t = __builtin_amdgcn_readlane(row, src), col_t = __builtin_amdgcn_readlane(col, src);
        const int64_t pi_t = ((int64_t)__builtin_amdgcn_readlane(pi_hi, src) << 32) | (unsigned)__builtin_amdgcn_readlane(pi_lo, src);
        const int pt_t = __builtin_amdgcn_readlane(pt_of_slot, src);
        tx.load(arena + (size_t)pi_t * patch_elems, a.H, a.W, row_t, col_t);
        if (a.v.d_refs && 2 * lane < C) rr = *reinterpret_cast<const double2*>(a.v.d_refs + (size_t)pt_t * C + 2 * lane);
        return src;
      };
      auto build = [&](const GramTexels<ST, C>& tx, const double2& rr, int src) {
        __syncthreads();                                   // (the previous build's reads of refbuf)
        if (2 * lane < C) *reinterpret_cast<double2*>(&lds.refbuf[2 * lane]) = rr;
        __syncthreads();
        gram_contract<ST, C>(tx, lds.refbuf, lds.G[src >> 2], lds.D[src >> 2]);
      };
      GramTexels<ST, C> t0, t1, t2;
      int s0 = take(t0, rr0), s1 = take(t1, rr1), s2 = take(t2, rr2);
      while (true) {
        if (s0 < 0) break;
        build(t0, rr0, s0); s0 = take(t0, rr0);
        if (s1 < 0) break;
        build(t1, rr1, s1); s1 = take(t1, rr1);
        if (s2 < 0) break;
        build(t2, rr2, s2); s2 = take(t2, rr2);
      }
      if (need && sub == 0) { ob[30] = (double)row; ob[31] = (double)col; }
      __syncthreads();                                     // the Gram matrices written by all lanes -> visible to their readers
      if (a.gram_G) {
        // ... and kept for the next call (and for the LM loop's evaluation, pxr_ba_gram.hip): an observation belongs to one
        // point, a point to one wavefront -- nobody else touches these 1 408 bytes
        unsigned long long wb = rebuilt;
        while (wb != 0ull) {
          const int src = __ffsll((long long)wb) - 1;
          wb &= ~(0xfull << (src & ~3));
          const int slot = src >> 2;
          const int64_t oi = slots[slot].obs;
          double2* g = reinterpret_cast<double2*>(a.gram_G + (size_t)oi * (IG_GDOUBLES + 16));
          const double2* Gd = reinterpret_cast<const double2*>(lds.G[slot]);
          g[lane] = Gd[lane];
          if (lane < 16) g[64 + lane] = Gd[64 + lane];
          else if (lane < 24) g[64 + lane] = reinterpret_cast<const double2*>(lds.D[slot])[lane - 16];
          const int2 cl = make_int2(__builtin_amdgcn_readlane(row, src), __builtin_amdgcn_readlane(col, src));
          if (lane == 0) a.gram_cell[oi] = cl;
        }
      }
    }
    if (act) {
      double wu[4], dwu[4], wv[4], dwv[4];
      catmull_rom_weights(u - cf, wu, dwu);
      catmull_rom_weights(v - rf, wv, dwv);
      const int R = sub;                                   // this lane's block row
      const double wvR = pick4(wv, R), dwvR = pick4(dwv, R);
      double Sgg = 0, Sgc = 0, Sgr = 0, Scc = 0, Scr = 0, Srr = 0;
      const double* Gq = lds.G[q];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const bool T = cb < R;                             // below the diagonal: the stored block (cb, R), transposed
        const double* blk = Gq + (T ? gram_block(cb, R) : gram_block(R, cb)) * 16;
        double e[16];
#pragma unroll
        for (int m = 0; m < 8; ++m) { const double2 t2 = *reinterpret_cast<const double2*>(blk + 2 * m); e[2 * m] = t2.x; e[2 * m + 1] = t2.y; }
        // forms on the STORED block: zw[c] = sum_r wu[r] e[r][c], zd[c] = sum_r dwu[r] e[r][c]
        double zw[4], zd[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          zw[c] = fma(wu[3], e[12 + c], fma(wu[2], e[8 + c], fma(wu[1], e[4 + c], wu[0] * e[c])));
          zd[c] = fma(dwu[3], e[12 + c], fma(dwu[2], e[8 + c], fma(dwu[1], e[4 + c], dwu[0] * e[c])));
        }
        const double Bww = fma(zw[3], wu[3], fma(zw[2], wu[2], fma(zw[1], wu[1], zw[0] * wu[0])));
        const double Bwd_s = fma(zw[3], dwu[3], fma(zw[2], dwu[2], fma(zw[1], dwu[1], zw[0] * dwu[0])));   // stored: rows wu, columns dwu
        const double Bdw_s = fma(zd[3], wu[3], fma(zd[2], wu[2], fma(zd[1], wu[1], zd[0] * wu[0])));
        const double Bdd = fma(zd[3], dwu[3], fma(zd[2], dwu[2], fma(zd[1], dwu[1], zd[0] * dwu[0])));
        const double Bwd = T ? Bdw_s : Bwd_s, Bdw = T ? Bwd_s : Bdw_s;          // B(a on MY rows, b on the block's columns)
        const double wvC = wv[cb], dwvC = dwv[cb];
        Sgg = fma(wvR * wvC, Bww, Sgg);
        Sgc = fma(wvR * wvC, Bwd, Sgc);
        Sgr = fma(wvR * dwvC, Bww, Sgr);
        Scc = fma(wvR * wvC, Bdd, Scc);
        Scr = fma(wvR * dwvC, Bdw, Scr);
        Srr = fma(dwvR * dwvC, Bww, Srr);
      }
      const double* Dq = lds.D[q] + 4 * R;
      const double2 d01 = *reinterpret_cast<const double2*>(Dq), d23 = *reinterpret_cast<const double2*>(Dq + 2);
      const double dw = fma(wu[3], d23.y, fma(wu[2], d23.x, fma(wu[1], d01.y, wu[0] * d01.x)));
      const double dd = fma(dwu[3], d23.y, fma(dwu[2], d23.x, fma(dwu[1], d01.y, dwu[0] * d01.x)));
      double Sfd = wvR * dw, Scd = wvR * dd, Srd = dwvR * dw;
      Sgg = quad_sum(Sgg); Sgc = quad_sum(Sgc); Sgr = quad_sum(Sgr);
      Scc = quad_sum(Scc); Scr = quad_sum(Scr); Srr = quad_sum(Srr);
      Sfd = quad_sum(Sfd); Scd = quad_sum(Scd); Srd = quad_sum(Srd);
      double s, gcc, gcr, grr, bc, br;
      if (l2) {
        const double ninv = inner_rsqrt(Sgg), n2inv = ninv * ninv;
        const double pc = Sgc * n2inv, pr = Sgr * n2inv;
        s = fmax(0.0, 1.0 - 2.0 * Sfd * ninv + lds.own[jq].r2);
        gcc = (Scc - Sgc * pc) * n2inv; gcr = (Scr - Sgc * pr) * n2inv; grr = (Srr - Sgr * pr) * n2inv;
        bc = -(Scd - Sfd * pc) * ninv; br = -(Srd - Sfd * pr) * ninv;
      } else {          // r = f - d
        s = fmax(0.0, Sgg - 2.0 * Sfd + lds.own[jq].r2);
        gcc = Scc; gcr = Scr; grr = Srr; bc = Sgc - Scd; br = Sgr - Srd;
      }
      double rho[3];
      inner_loss(a.loss.type, a.loss.a, s, rho);
      double cq = 0.5 * rho[0];
      if (a.check_bounds && !a.v.d_refs && !(u > 0.0 && u < (double)a.W && v > 0.0 && v < (double)a.H)) cq = __builtin_nan("");
      gcc *= sx * sx; gcr *= sx * sy; grr *= sy * sy; bc *= sx; br *= sy;
      double kappa = 0.0;   // Ceres' corrector (corrector.cc): alpha = 1 - sqrt(1 + 2 s rho'' / rho')
      if (s != 0.0 && rho[2] > 0.0) {
        const double Dc = 1.0 + 2.0 * s * rho[2] * inner_rcp(rho[1]);
        const double alpha = 1.0 - sqrt(Dc);
        kappa = (2.0 * alpha - alpha * alpha) * inner_rcp(s);
      }
      const double w8 = rho[1];
      const double m00 = w8 * (gcc - kappa * bc * bc), m01 = w8 * (gcr - kappa * bc * br), m11 = w8 * (grr - kappa * br * br);
      const double b0 = w8 * bc, b1 = w8 * br;
      double me0[3], me1[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) { me0[m] = m00 * PX[0][m] + m01 * PX[1][m]; me1[m] = m01 * PX[0][m] + m11 * PX[1][m]; }
      if (sub == 0) {
        double* rs = lds.res[q];
        rs[0] = cq;
        rs[1] = PX[0][0] * me0[0] + PX[1][0] * me1[0];
        rs[2] = PX[0][0] * me0[1] + PX[1][0] * me1[1];
        rs[3] = PX[0][0] * me0[2] + PX[1][0] * me1[2];
        rs[4] = PX[0][1] * me0[1] + PX[1][1] * me1[1];
        rs[5] = PX[0][1] * me0[2] + PX[1][1] * me1[2];
        rs[6] = PX[0][2] * me0[2] + PX[1][2] * me1[2];
#pragma unroll
        for (int m = 0; m < 3; ++m) rs[7 + m] = PX[0][m] * b0 + PX[1][m] * b1;
      }
    }
    __syncthreads();
    // -- owners: the sum over their slots in track order, then Ceres' trust-region bookkeeping on the state in LDS --
    double cand = 0.0, Hc[6] = {0, 0, 0, 0, 0, 0}, gc[3] = {0, 0, 0};
    if (owner && lds.own[lane].live != 0) {
      for (int slq = my_b; slq < my_e; ++slq) {
        const double* rs = lds.res[slq];
        cand += rs[0];
#pragma unroll
        for (int m = 0; m < 6; ++m) Hc[m] += rs[1 + m];
#pragma unroll
        for (int m = 0; m < 3; ++m) gc[m] += rs[7 + m];
      }
    }
    if (owner) inner_owner_update(lds.own[lane], cand, Hc, gc, first, my_e > my_b, active, a.cost_before, a.cost_pt ? a.cost_pt + myp : nullptr);
    first = false;
    if (__ballot(active) == 0) break;
  }
  if (variable) {
    const InnerOwner& S = lds.own[lane];
    a.xyz_out[3 * myp] = S.X[0]; a.xyz_out[3 * myp + 1] = S.X[1]; a.xyz_out[3 * myp + 2] = S.X[2];
  }
}

// Enqueue the inner iterations on the candidate parameters `view` (xyz refined in place);
// *d_cost_before (device double, caller-zeroed) receives the cost at the unrefined candidate.
// lists (may be NULL): the points with at most IG_MAXO observations (Gram-matrix kernel) and the others (packed kernel), made
// once per solve by make_inner_lists.
int launch_inner_iterations(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                            const pxr_loss* loss, const int64_t* d_pt_ptr, const int64_t* d_pt_obs,
                            const int* d_pt_var, double* d_cost_before, const InnerLists* lists, double* d_cost_per_point,
                            const GramCache* gram, bool gram_warm) {
  if (arena->C != 128 && arena->C != 64 && arena->C != 3 && arena->C != 1)
    return set_error(PXR_EUNSUPPORTED, "inner iterations: CHANNELS=%d not supported (128, 64; cost maps: 3, 1)", arena->C);
  InnerArgs a;
  a.v = *view;
  a.arena = arena->d_data; a.corners = arena->d_corners; a.scales = arena->d_scales;
  a.H = arena->H; a.W = arena->W; a.up = arena->up; a.l2_normalize = cfg->l2_normalize; a.check_bounds = cfg->check_bounds; a.loss = *loss;
  a.pt_ptr = d_pt_ptr; a.pt_obs = d_pt_obs; a.pt_var = d_pt_var;
  a.xyz_out = const_cast<double*>(view->d_xyz); a.cost_before = d_cost_before; a.cost_pt = d_cost_per_point;
  a.gram_G = gram ? gram->G : nullptr; a.gram_cell = gram ? static_cast<int2*>(gram->cell) : nullptr; a.gram_warm = gram_warm ? 1 : 0;
  const int ppb = arena->C >= 64 ? 1 : 8;   // points per workgroup (InnerShape)
  const int threads = 64;
  const unsigned blocks = (unsigned)((view->n_points + ppb - 1) / ppb);
  if (blocks == 0) return PXR_OK;
#define INNER_LAUNCH(KERNEL, ST, CC)                                                                          \
  do {                                                                                                        \
    if (cfg->use_float_simd) hipLaunchKernelGGL((KERNEL<ST, CC, true>), dim3(blocks), dim3(threads), 0, ctx->stream, a);  \
    else hipLaunchKernelGGL((KERNEL<ST, CC, false>), dim3(blocks), dim3(threads), 0, ctx->stream, a);             \
  } while (0)
#define INNER_PACKED(ST, CC, NBLK, PPW, LIST)                                                                             \
  do {                                                                                                                    \
    if (cfg->use_float_simd) hipLaunchKernelGGL((k_inner_packed<ST, CC, true, 4>), dim3(NBLK), dim3(64), 0, ctx->stream, a, PPW, LIST);  \
    else hipLaunchKernelGGL((k_inner_packed<ST, CC, false, 4>), dim3(NBLK), dim3(64), 0, ctx->stream, a, PPW, LIST);         \
  } while (0)
#define INNER_BY_STORAGE(MACRO, ...)                                                    \
  do {                                                                                  \
    if (arena->dtype == PXR_F16 && arena->C == 128) MACRO(_Float16, 128, __VA_ARGS__);  \
    else if (arena->dtype == PXR_F16) MACRO(_Float16, 64, __VA_ARGS__);                 \
    else if (arena->C == 128) MACRO(float, 128, __VA_ARGS__);                           \
    else MACRO(float, 64, __VA_ARGS__);                                                 \
  } while (0)
#define INNER_GRAM(ST, CC, NBLK, MAXO)                                                                                    \
  hipLaunchKernelGGL((k_inner_gram<ST, CC>), dim3(NBLK), dim3(64), sizeof(double) * gram_lds_doubles(MAXO, CC), ctx->stream, a, \
                     static_cast<const char*>(lists->d_slots), MAXO)
#define INNER_GRAM_PACKED(ST, CC, NBLK) \
  hipLaunchKernelGGL((k_inner_gram_packed<ST, CC>), dim3(NBLK), dim3(64), 0, ctx->stream, a, static_cast<const char*>(lists->d_waves))
  if (arena->C <= 4) {
    if (arena->dtype == PXR_F16 && arena->C == 3) INNER_LAUNCH(k_inner_points, _Float16, 3);
    else if (arena->dtype == PXR_F16) INNER_LAUNCH(k_inner_points, _Float16, 1);
    else if (arena->dtype == PXR_F32 && arena->C == 3) INNER_LAUNCH(k_inner_points, float, 3);
    else if (arena->dtype == PXR_F32) INNER_LAUNCH(k_inner_points, float, 1);
    else if (arena->C == 3) INNER_LAUNCH(k_inner_points, double, 3);
    else INNER_LAUNCH(k_inner_points, double, 1);
  } else if (getenv("PXR_INNER_OLD")) {   // A/B knob (tools/fuzz_solve_vs_oracle.py): the one-point-per-wavefront kernel for every storage type
    if (arena->dtype == PXR_F16 && arena->C == 128) INNER_LAUNCH(k_inner_points, _Float16, 128);
    else if (arena->dtype == PXR_F16) INNER_LAUNCH(k_inner_points, _Float16, 64);
    else if (arena->dtype == PXR_F32 && arena->C == 128) INNER_LAUNCH(k_inner_points, float, 128);
    else if (arena->dtype == PXR_F32) INNER_LAUNCH(k_inner_points, float, 64);
    else if (arena->C == 128) INNER_LAUNCH(k_inner_points, double, 128);
    else INNER_LAUNCH(k_inner_points, double, 64);
  } else if (arena->dtype != PXR_F64) {
    // PXR_INNER_PACKED: A/B knob -- the round-3 kernel (descriptor interpolated at every round) for every point
    if (lists == nullptr || getenv("PXR_INNER_PACKED")) {
      // packed kernel: points per wavefront from the mean track length (16 observation slots per trip)
      const int64_t per16 = view->n_obs > 0 ? (16 * view->n_points) / view->n_obs : 1;
      const int ppw = (int)(per16 < 1 ? 1 : (per16 > IP_MAXPTS ? IP_MAXPTS : per16));
      const unsigned pblocks = (unsigned)((view->n_points + ppw - 1) / ppw);
      const int* no_list = nullptr;
      INNER_BY_STORAGE(INNER_PACKED, pblocks, ppw, no_list);
    } else {
      // Gram-matrix kernel for the points whose observations' Gram matrices fit a wavefront's LDS, packed kernel (one point per
      // wavefront) for the long tracks
      // short tracks: up to four points (sixteen observation slots) per wavefront in lockstep; PXR_INNER_GRAM1=1 is the A/B knob
      // for the one-point-per-wavefront kernel
      if (lists->n_waves > 0) {
        if (gram && !getenv("PXR_INNER_NO_PREBUILD")) {
          // the matrices the kernel will find stale are rebuilt before it starts (k_gram_flag_slots); after that the cache is warm
          PXR_HIP(hipMemsetAsync(gram->list, 0, sizeof(int) * (size_t)view->n_obs, ctx->stream));
          const int64_t n_thr = lists->n_waves * GW_SLOTS;
          hipLaunchKernelGGL(k_gram_flag_slots, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, ctx->stream, a, static_cast<const char*>(lists->d_waves),
                             lists->n_waves, gram->list);
          if (int rc = gram_build_flagged(ctx, arena, view, *gram)) return rc;
          a.gram_warm = 1;
        }
        INNER_BY_STORAGE(INNER_GRAM_PACKED, (unsigned)lists->n_waves);
      }
      else if (lists->n_short > 0 && lists->d_slots) INNER_BY_STORAGE(INNER_GRAM, (unsigned)lists->n_short, lists->maxo_short);
      if (lists->n_long > 0) INNER_BY_STORAGE(INNER_PACKED, (unsigned)lists->n_long, 1, lists->d_long);
    }
  } else if (arena->dtype == PXR_F64 && arena->C == 128) INNER_LAUNCH(k_inner_points, double, 128);
  else INNER_LAUNCH(k_inner_points, double, 64);
#undef INNER_LAUNCH
#undef INNER_PACKED
#undef INNER_GRAM
#undef INNER_GRAM_PACKED
#undef INNER_BY_STORAGE
  return hip_check(hipGetLastError(), "k_inner_points launch");
}

// The host's split of the points for the inner iterations: track lengths from the CSR of the point-ordered observation list.
size_t inner_wave_stage_bytes(int64_t n_pts) { return sizeof(GramWave) * (size_t)(n_pts + 1); }
// h_wave_stage (optional): pinned host memory of inner_wave_stage_bytes(n) bytes the wave headers are packed INTO, so that their
// upload is one DMA from where they were built (3.4 MB at 200k points: 0.3 ms through a pageable vector)
int make_inner_lists(pxr_ctx* ctx, const int64_t* pt_ptr, int64_t n, void* h_wave_stage, size_t wave_stage_bytes, const pxr_ba_view* view,
                     const int64_t* d_pt_ptr, const int64_t* d_pt_obs, InnerLists* out) {
  hipStream_t st = ctx->stream;
  const bool one_point_kernel = getenv("PXR_INNER_GRAM1") != nullptr;      // (A/B knob: k_inner_gram's tables instead of the packed kernel's)
  std::vector<GramPoint> shorts;
  std::vector<GramWave> waves_pageable;
  GramWave* wv = wave_stage_bytes >= inner_wave_stage_bytes(n) ? static_cast<GramWave*>(h_wave_stage) : nullptr;
  int64_t n_wv = 0;
  std::vector<int> longs;
  int maxo = 1;
  const int max_pts = getenv("PXR_INNER_GRAM_PPW") ? std::max(1, std::min(IP_MAXPTS, atoi(getenv("PXR_INNER_GRAM_PPW")))) : IP_MAXPTS;   // (A/B knob)
  if (one_point_kernel) shorts.reserve((size_t)n); else if (!wv) waves_pageable.reserve((size_t)n / 2 + 1);
  int64_t n_short = 0;
  GramWave cur{};                                           // consecutive short points are packed: <= 4 points, <= 16 observations
  auto flush = [&]() {
    if (cur.npts > 0) { if (wv) wv[n_wv] = cur; else waves_pageable.push_back(cur); ++n_wv; }
    cur = GramWave{};
  };
  for (int64_t p = 0; p < n; ++p) {
    const int64_t len = pt_ptr[p + 1] - pt_ptr[p];
    if (len <= 0) continue;
    if (len <= IG_MAXO) {
      ++n_short; maxo = std::max(maxo, (int)len);
      if (one_point_kernel) { shorts.push_back(GramPoint{(int)p, (int)len, pt_ptr[p]}); continue; }
      if (cur.npts == max_pts || cur.st[cur.npts] + (int)len > GW_SLOTS) flush();
      cur.p[cur.npts] = (int)p;
      cur.st[cur.npts + 1] = cur.st[cur.npts] + (int)len;
      ++cur.npts;
    } else longs.push_back((int)p);
  }
  flush();
  const GramWave* const waves = wv ? wv : waves_pageable.data();
  out->n_waves = n_wv; out->d_waves = nullptr; out->d_wave_heads = nullptr;
  out->n_short = n_short; out->n_long = (int64_t)longs.size(); out->maxo_short = maxo;
  out->d_short = nullptr; out->d_long = nullptr; out->d_slots = nullptr;
  if (!shorts.empty()) {
    out->d_short = solve_scratch(sizeof(GramPoint) * shorts.size(), &out->own_short);
    if (!out->d_short) return set_error(PXR_ENOMEM, "hipMalloc(inner lists)");
    if (int rc = hip_check(hipMemcpyAsync(out->d_short, shorts.data(), sizeof(GramPoint) * shorts.size(), hipMemcpyHostToDevice, st), "H2D")) return rc;
    out->d_slots = solve_scratch(gram_entry_bytes(maxo) * shorts.size(), &out->own_slots);
    if (!out->d_slots) return set_error(PXR_ENOMEM, "hipMalloc(inner table)");
    const int64_t n_thr = (int64_t)shorts.size() * (maxo + 1);
    hipLaunchKernelGGL(k_gram_table, dim3((unsigned)((n_thr + 255) / 256)), dim3(256), 0, st, (int64_t)shorts.size(), maxo,
                       static_cast<const GramPoint*>(out->d_short), d_pt_obs, view->d_obs_image, view->d_obs_patch, view->d_image_camera,
                       static_cast<char*>(out->d_slots));
  }
  if (n_wv > 0) {            // one allocation: the headers as the host packed them, then the table the kernel reads
    const size_t heads = (sizeof(GramWave) * (size_t)n_wv + 255) & ~(size_t)255;
    char* d_all = static_cast<char*>(solve_scratch(heads + gram_wave_bytes() * (size_t)n_wv, &out->own_heads));
    if (!d_all) return set_error(PXR_ENOMEM, "hipMalloc(inner wave table)");
    out->d_wave_heads = d_all; out->d_waves = d_all + heads;
    if (int rc = hip_check(hipMemcpyAsync(d_all, waves, sizeof(GramWave) * (size_t)n_wv, hipMemcpyHostToDevice, st), "H2D")) return rc;
    const int64_t n_thr2 = n_wv * (GW_SLOTS + 1);
    hipLaunchKernelGGL(k_gram_wave_table, dim3((unsigned)((n_thr2 + 255) / 256)), dim3(256), 0, st, n_wv,
                       reinterpret_cast<const GramWave*>(d_all), d_pt_ptr, d_pt_obs, view->d_obs_image, view->d_obs_patch, view->d_image_camera,
                       static_cast<char*>(out->d_waves));
  }
  if (!longs.empty()) {
    out->d_long = static_cast<int*>(solve_scratch(sizeof(int) * longs.size(), &out->own_long));
    if (!out->d_long) return set_error(PXR_ENOMEM, "hipMalloc(inner lists)");
    if (int rc = hip_check(hipMemcpyAsync(out->d_long, longs.data(), sizeof(int) * longs.size(), hipMemcpyHostToDevice, st), "H2D")) return rc;
  }
  return hip_check(hipStreamSynchronize(st), "inner lists upload");    // the host vectors go out of scope
}
void free_inner_lists(InnerLists* l) {
  if (l->d_short && l->own_short) (void)hipFree(l->d_short);
  if (l->d_long && l->own_long) (void)hipFree(l->d_long);
  if (l->d_slots && l->own_slots) (void)hipFree(l->d_slots);
  if (l->d_wave_heads && l->own_heads) (void)hipFree(l->d_wave_heads);      // (d_waves lies inside it)
  l->d_short = nullptr; l->d_long = nullptr; l->d_slots = nullptr; l->d_waves = nullptr; l->d_wave_heads = nullptr;
}

}  // namespace pxr

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
// Mapping: one wave per point, its rows of C/8 lanes (four at C = 128, eight at C = 64) evaluate that many
// observations at a time with
// the same interpolation core as the fused BA kernel; the per-observation 2x2 / 2-vector blocks
// are folded with d(x,y)/dX into the point's 3x3 normal matrix and reduced across the rows with
// two shuffles.  The whole nested LM runs in registers; consecutive evaluations re-read the same
// 4 KiB stencils from L2.  The kernel also returns the cost at the (unrefined) candidate, so the
// outer loop needs no separate evaluation for Ceres' inner-iteration bookkeeping.
#include <hip/hip_runtime.h>

#include "pxr_device.h"
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
  if (lane == 0) atomicAdd(a.cost_before, cost);
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

// Two entry points over the same body: the fp16 / fp32 instantiations are capped at 256 VGPRs (two
// wavefronts per SIMD; unconstrained they take ~330 and run one), the fp64-storage ones keep the
// compiler's budget (capped they would spill several hundred registers).
template <typename ST, int C, bool FS>
__global__ __launch_bounds__(64 * InnerShape<C>::WPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_inner_points_occ2(const InnerArgs a) {
  __shared__ double sh_obs[InnerShape<C>::WPB * InnerShape<C>::PPW][InnerShape<C>::MAXO][26];   // per point: q(4) t(3) k(12) sx sy corner(2) model patch
  inner_points_body<ST, C, FS>(a, sh_obs);
}
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
// -- the same quantities as PixelInterpolator's normalise-then-subtract (interpolation.h:648-666) up to rounding
// (1e-13 relative on r.r at the residual sizes of a converging problem; r.r is clamped at 0).
constexpr int IP_SLOTS = 16, IP_MAXPTS = 4, IP_MAXQ = 32, IP_OBS = 26;

template <int C>
struct PackedLds {
  double ref[IP_MAXPTS][C];          // reference descriptors of the wavefront's points
  double obs[IP_MAXQ][IP_OBS];       // per observation: q(4) t(3) k(12) sx sy corner(2) model patch
  double res[IP_SLOTS][10];          // per slot of the current trip: cost, H (6), g (3)
  double Xc[IP_MAXPTS][4];           // candidate position per point; [3] = d.d of its reference
  double cost0[IP_MAXPTS];
  int live[IP_MAXPTS];               // point still iterating (or first round): its slots are evaluated
};

__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  return v;
}

template <typename ST, int C, bool FS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_inner_packed(const InnerArgs a, const int ppw) {
  static_assert(C == 128 || C == 64, "four lanes per observation, C / 32 chunks of 8 channels per lane");
  constexpr int NCHUNK = C / 32;
  __shared__ PackedLds<C> lds;
  const int lane = threadIdx.x, sidx = lane >> 2, sub = lane & 3;
  const int64_t P0 = (int64_t)blockIdx.x * ppw;
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
  bool active = owner && my_e > my_b && a.pt_var[myp] != 0;
  double X[3] = {a.v.d_xyz[3 * myp], a.v.d_xyz[3 * myp + 1], a.v.d_xyz[3 * myp + 2]};
  const ST* arena = reinterpret_cast<const ST*>(a.arena);
  const size_t patch_elems = (size_t)a.H * a.W * C;
  const bool l2 = a.l2_normalize != 0;

  // ---- staging: references, observation records ----
  for (int j = 0; j < npts; ++j)
    for (int ch = lane; ch < C; ch += 64) lds.ref[j][ch] = a.v.d_refs ? a.v.d_refs[(size_t)(P0 + j) * C + ch] : 0.0;
  for (int q = sidx; q < L && q < IP_MAXQ; q += IP_SLOTS) {
    double* ob = lds.obs[q];
    const int64_t i = a.pt_obs[o0 + q];
    const int img = a.v.d_obs_image[i], cam = a.v.d_image_camera[img];
    const int64_t pi = a.v.d_obs_patch[i];
    if (sub == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) ob[j] = a.v.d_qvec[4 * (size_t)img + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) ob[4 + j] = a.v.d_tvec[3 * (size_t)img + j];
    } else if (sub == 1) {
#pragma unroll
      for (int j = 0; j < 6; ++j) ob[7 + j] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + j];
    } else if (sub == 2) {
#pragma unroll
      for (int j = 6; j < PXR_KPAD; ++j) ob[7 + j] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + j];
    } else {
      ob[19] = a.scales[2 * pi]; ob[20] = a.scales[2 * pi + 1];
      ob[21] = (double)a.corners[2 * pi]; ob[22] = (double)a.corners[2 * pi + 1];
      ob[23] = (double)a.v.d_cam_model[cam]; ob[24] = (double)pi;
    }
  }
  __syncthreads();
  if (owner) {
    double r2 = 0.0;
    if (l2)
      for (int ch = 0; ch < C; ++ch) r2 = fma(lds.ref[lane][ch], lds.ref[lane][ch], r2);
    lds.Xc[lane][3] = r2;
  }

  // ---- the nested LMs of the wavefront's points, one evaluation (all points) per round ----
  double cost = 0.0, H[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, sc[3] = {1, 1, 1};
  double radius = 1e4, decrease_factor = 2.0, diag[3] = {0, 0, 0}, mcc = 0.0;
  double Xc[3] = {X[0], X[1], X[2]};
  int invalid = 0, it = 0;
  bool reuse_diag = false, first = true;
  while (true) {
    // a point that has finished keeps its slots but they skip the evaluation: the wavefront's rounds are those of its slowest
    // point, the texel traffic (the kernel moves ~4 KB per observation and round through an L2 it does not fit) only that of
    // the points still iterating
    if (owner) { lds.Xc[lane][0] = Xc[0]; lds.Xc[lane][1] = Xc[1]; lds.Xc[lane][2] = Xc[2]; lds.live[lane] = (first || active) ? 1 : 0; }
    __syncthreads();
    // -- evaluation: cost, H (xx xy xz yy yz zz), g of every point at its candidate --
    double cand = 0.0, Hc[6] = {0, 0, 0, 0, 0, 0}, gc[3] = {0, 0, 0};
    for (int base = 0; base < L; base += IP_SLOTS) {
      const int q = base + sidx;
      const bool valid = q < L;
      const int qc = valid ? q : L - 1;
      const int j = (qc >= st1) + (qc >= st2) + (qc >= st3);
      if (valid && lds.live[j]) {
      double qv[4], t[3], k[PXR_KPAD], sx, sy, cx, cy;
      int model;
      int64_t pi;
      if (qc < IP_MAXQ) {
        const double* ob = lds.obs[qc];
#pragma unroll
        for (int m = 0; m < 4; ++m) qv[m] = ob[m];
#pragma unroll
        for (int m = 0; m < 3; ++m) t[m] = ob[4 + m];
#pragma unroll
        for (int m = 0; m < PXR_KPAD; ++m) k[m] = ob[7 + m];
        sx = ob[19]; sy = ob[20]; cx = ob[21]; cy = ob[22];
        model = (int)ob[23]; pi = (int64_t)ob[24];
      } else {   // long tracks: the tail of the list comes from global memory
        const int64_t i = a.pt_obs[o0 + qc];
        const int img = a.v.d_obs_image[i], cam = a.v.d_image_camera[img];
        pi = a.v.d_obs_patch[i];
#pragma unroll
        for (int m = 0; m < 4; ++m) qv[m] = a.v.d_qvec[4 * (size_t)img + m];
#pragma unroll
        for (int m = 0; m < 3; ++m) t[m] = a.v.d_tvec[3 * (size_t)img + m];
#pragma unroll
        for (int m = 0; m < PXR_KPAD; ++m) k[m] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + m];
        model = a.v.d_cam_model[cam];
        sx = a.scales[2 * pi]; sy = a.scales[2 * pi + 1];
        cx = (double)a.corners[2 * pi]; cy = (double)a.corners[2 * pi + 1];
      }
      const double Xs[3] = {lds.Xc[j][0], lds.Xc[j][1], lds.Xc[j][2]};
      double x, y, A[2][3], Pq[2][4], PX[2][3], Pk[2][PXR_KPAD];
      world_to_pixel_jac(model, k, qv, t, Xs, x, y, A, Pq, PX, Pk);
      const double u = x * sx - 0.5 - cx, v = y * sy - 0.5 - cy;   // FeaturePatch::ToPixelCoordinates (featurepatch.h:250-255)
      const ST* patch = arena + (size_t)pi * patch_elems;
      // sums over this lane's channels: g = f (normalised mode) or f - d
      double Sgg = 0, Sgc = 0, Sgr = 0, Scc = 0, Scr = 0, Srr = 0, Sfd = 0, Scd = 0, Srd = 0;
#pragma unroll 1
      for (int c = 0; c < NCHUNK; ++c) {
        const int chan0 = (c * 4 + sub) * 8;
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
      Sgg = quad_sum(Sgg); Sgc = quad_sum(Sgc); Sgr = quad_sum(Sgr);
      Scc = quad_sum(Scc); Scr = quad_sum(Scr); Srr = quad_sum(Srr);
      double s, gcc, gcr, grr, bc, br;
      if (l2) {
        Sfd = quad_sum(Sfd); Scd = quad_sum(Scd); Srd = quad_sum(Srd);
        const double ninv = 1.0 / sqrt(Sgg), n2inv = ninv * ninv;
        const double pc = Sgc * n2inv, pr = Sgr * n2inv;
        s = fmax(0.0, 1.0 - 2.0 * Sfd * ninv + lds.Xc[j][3]);
        gcc = (Scc - Sgc * pc) * n2inv; gcr = (Scr - Sgc * pr) * n2inv; grr = (Srr - Sgr * pr) * n2inv;
        bc = -(Scd - Sfd * pc) * ninv; br = -(Srd - Sfd * pr) * ninv;
      } else {
        s = Sgg; gcc = Scc; gcr = Scr; grr = Srr; bc = Sgc; br = Sgr;
      }
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, s, rho);
      double cq = 0.5 * rho[0];
      // check_bounds: without a reference descriptor a projection outside its patch fails the evaluation
      // (feature_reference.h:128-130) -> non-finite cost -> the step is rejected
      if (a.check_bounds && !a.v.d_refs && !(u > 0.0 && u < (double)a.W && v > 0.0 && v < (double)a.H)) cq = __builtin_nan("");
      gcc *= sx * sx; gcr *= sx * sy; grr *= sy * sy; bc *= sx; br *= sy;
      double kappa = 0.0;
      if (s != 0.0 && rho[2] > 0.0) {
        const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
        const double alpha = 1.0 - sqrt(D);
        kappa = (2.0 * alpha - alpha * alpha) / s;
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
      __syncthreads();
    }
    // -- owners: Ceres' trust-region bookkeeping (same loop as k_inner_points above) --
    if (first) {
      first = false;
      cost = cand;
#pragma unroll
      for (int m = 0; m < 6; ++m) H[m] = Hc[m];
#pragma unroll
      for (int m = 0; m < 3; ++m) g[m] = gc[m];
      if (owner) lds.cost0[lane] = my_e > my_b ? cost : 0.0;
      __syncthreads();
      if (lane == 0) {
        double c0 = 0.0;
        for (int j = 0; j < npts; ++j) c0 += lds.cost0[j];
        atomicAdd(a.cost_before, c0);
      }
      if (active) {
        const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        if (gmax <= 1e-10) active = false;
        sc[0] = 1.0 / (1.0 + sqrt(H[0])); sc[1] = 1.0 / (1.0 + sqrt(H[3])); sc[2] = 1.0 / (1.0 + sqrt(H[5]));
        H[0] *= sc[0] * sc[0]; H[1] *= sc[0] * sc[1]; H[2] *= sc[0] * sc[2];
        H[3] *= sc[1] * sc[1]; H[4] *= sc[1] * sc[2]; H[5] *= sc[2] * sc[2];
        g[0] *= sc[0]; g[1] *= sc[1]; g[2] *= sc[2];
      }
    } else if (active) {
      const double s2 = (Xc[0] - X[0]) * (Xc[0] - X[0]) + (Xc[1] - X[1]) * (Xc[1] - X[1]) + (Xc[2] - X[2]) * (Xc[2] - X[2]);
      const double x2 = X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
      const double cost_change = cost - cand;
      if (sqrt(s2) <= 1e-8 * (sqrt(x2) + 1e-8)) active = false;
      else if (fabs(cost_change) <= 1e-6 * cost) active = false;
      else {
        const double rel = cost_change / mcc;
        if (rel > 1e-3) {
          X[0] = Xc[0]; X[1] = Xc[1]; X[2] = Xc[2];
          cost = cand;
#pragma unroll
          for (int m = 0; m < 6; ++m) H[m] = Hc[m];
#pragma unroll
          for (int m = 0; m < 3; ++m) g[m] = gc[m];
          const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
          H[0] *= sc[0] * sc[0]; H[1] *= sc[0] * sc[1]; H[2] *= sc[0] * sc[2];
          H[3] *= sc[1] * sc[1]; H[4] *= sc[1] * sc[2]; H[5] *= sc[2] * sc[2];
          g[0] *= sc[0]; g[1] *= sc[1]; g[2] *= sc[2];
          const double tmp = 2.0 * rel - 1.0;
          radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp));
          decrease_factor = 2.0; reuse_diag = false;
          if (gmax <= 1e-10) active = false;
        } else {
          radius /= decrease_factor; decrease_factor *= 2.0; reuse_diag = true;
        }
      }
    }
    // -- owners: the next candidate (steps that fail without an evaluation are retried here, like the `continue` above) --
    while (active) {
      if (it >= 50 || radius < 1e-32) { active = false; break; }
      ++it;
      if (!reuse_diag) {
        diag[0] = fmin(fmax(H[0], 1e-6), 1e32); diag[1] = fmin(fmax(H[3], 1e-6), 1e32); diag[2] = fmin(fmax(H[5], 1e-6), 1e32);
      }
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
      double stp[3];
      stp[2] = y2 / l22; stp[1] = (y1 - l21 * stp[2]) / l11; stp[0] = (y0 - l10 * stp[1] - l20 * stp[2]) / l00;
      if (ok) {
        const double dg = stp[0] * g[0] + stp[1] * g[1] + stp[2] * g[2];
        const double dHd = stp[0] * (H[0] * stp[0] + H[1] * stp[1] + H[2] * stp[2]) + stp[1] * (H[1] * stp[0] + H[3] * stp[1] + H[4] * stp[2]) +
                           stp[2] * (H[2] * stp[0] + H[4] * stp[1] + H[5] * stp[2]);
        mcc = -dg - 0.5 * dHd;
        if (!(mcc > 0.0) || !isfinite(stp[0]) || !isfinite(stp[1]) || !isfinite(stp[2])) ok = false;
      }
      if (!ok) {
        if (++invalid >= 5) { active = false; break; }
        radius *= 0.5; reuse_diag = true;
        continue;
      }
      invalid = 0;
      Xc[0] = X[0] + stp[0] * sc[0]; Xc[1] = X[1] + stp[1] * sc[1]; Xc[2] = X[2] + stp[2] * sc[2];
      break;
    }
    if (__ballot(active) == 0) break;
  }
  if (owner && my_e > my_b && a.pt_var[myp] != 0) { a.xyz_out[3 * myp] = X[0]; a.xyz_out[3 * myp + 1] = X[1]; a.xyz_out[3 * myp + 2] = X[2]; }
}

// Enqueue the inner iterations on the candidate parameters `view` (xyz refined in place);
// *d_cost_before (device double, caller-zeroed) receives the cost at the unrefined candidate.
int launch_inner_iterations(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                            const pxr_loss* loss, const int64_t* d_pt_ptr, const int64_t* d_pt_obs,
                            const int* d_pt_var, double* d_cost_before) {
  if (arena->C != 128 && arena->C != 64 && arena->C != 3 && arena->C != 1)
    return set_error(PXR_EUNSUPPORTED, "inner iterations: CHANNELS=%d not supported (128, 64; cost maps: 3, 1)", arena->C);
  InnerArgs a;
  a.v = *view;
  a.arena = arena->d_data; a.corners = arena->d_corners; a.scales = arena->d_scales;
  a.H = arena->H; a.W = arena->W; a.up = arena->up; a.l2_normalize = cfg->l2_normalize; a.check_bounds = cfg->check_bounds; a.loss = *loss;
  a.pt_ptr = d_pt_ptr; a.pt_obs = d_pt_obs; a.pt_var = d_pt_var;
  a.xyz_out = const_cast<double*>(view->d_xyz); a.cost_before = d_cost_before;
  const int ppb = arena->C >= 64 ? 1 : 8;   // points per workgroup (InnerShape)
  const int threads = 64;
  const unsigned blocks = (unsigned)((view->n_points + ppb - 1) / ppb);
  if (blocks == 0) return PXR_OK;
#define INNER_LAUNCH(KERNEL, ST, CC)                                                                          \
  do {                                                                                                        \
    if (cfg->use_float_simd) hipLaunchKernelGGL((KERNEL<ST, CC, true>), dim3(blocks), dim3(threads), 0, ctx->stream, a);  \
    else hipLaunchKernelGGL((KERNEL<ST, CC, false>), dim3(blocks), dim3(threads), 0, ctx->stream, a);             \
  } while (0)
  if (arena->C <= 4) {
    if (arena->dtype == PXR_F16 && arena->C == 3) INNER_LAUNCH(k_inner_points, _Float16, 3);
    else if (arena->dtype == PXR_F16) INNER_LAUNCH(k_inner_points, _Float16, 1);
    else if (arena->dtype == PXR_F32 && arena->C == 3) INNER_LAUNCH(k_inner_points, float, 3);
    else if (arena->dtype == PXR_F32) INNER_LAUNCH(k_inner_points, float, 1);
    else if (arena->C == 3) INNER_LAUNCH(k_inner_points, double, 3);
    else INNER_LAUNCH(k_inner_points, double, 1);
  } else if (arena->dtype != PXR_F64) {
    // packed kernel: points per wavefront from the mean track length (16 observation slots per trip)
    const int64_t per16 = view->n_obs > 0 ? (16 * view->n_points) / view->n_obs : 1;
    const int ppw = (int)(per16 < 1 ? 1 : (per16 > IP_MAXPTS ? IP_MAXPTS : per16));
    const unsigned pblocks = (unsigned)((view->n_points + ppw - 1) / ppw);
#define INNER_PACKED(ST, CC)                                                                                              \
  do {                                                                                                                    \
    if (cfg->use_float_simd) hipLaunchKernelGGL((k_inner_packed<ST, CC, true>), dim3(pblocks), dim3(64), 0, ctx->stream, a, ppw);  \
    else hipLaunchKernelGGL((k_inner_packed<ST, CC, false>), dim3(pblocks), dim3(64), 0, ctx->stream, a, ppw);           \
  } while (0)
    if (arena->dtype == PXR_F16 && arena->C == 128) INNER_PACKED(_Float16, 128);
    else if (arena->dtype == PXR_F16) INNER_PACKED(_Float16, 64);
    else if (arena->C == 128) INNER_PACKED(float, 128);
    else INNER_PACKED(float, 64);
#undef INNER_PACKED
  } else if (arena->dtype == PXR_F64 && arena->C == 128) INNER_LAUNCH(k_inner_points, double, 128);
  else INNER_LAUNCH(k_inner_points, double, 64);
#undef INNER_LAUNCH
  return hip_check(hipGetLastError(), "k_inner_points launch");
}

}  // namespace pxr

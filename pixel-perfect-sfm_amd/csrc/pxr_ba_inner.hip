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
template <int C> struct InnerShape { static constexpr int PPW = C >= 64 ? 1 : 8, MAXO = C >= 64 ? 16 : 8; };

template <typename ST, int C, bool FS>   // FS: InterpolationConfig.use_float_simd
__device__ __forceinline__ void inner_points_body(const InnerArgs& a, double (*sh_obs)[InnerShape<C>::MAXO][26]) {
  static_assert(C == 128 || C == 64 || C <= 4, "one observation per row of C / 8 lanes, or per lane (cost maps)");
  // features: one point per wavefront, an observation per row of C / 8 lanes.  Cost maps (C = 1, 3;
  // costmap_bundle_optimizer.h:9-14): the whole texel in one lane, so a point takes 8 lanes (one DPP half-row: tracks
  // of up to 8 observations in one pass) and a wavefront runs 8 points, each group its own nested LM
  constexpr int PPW = InnerShape<C>::PPW, INNER_MAXO = InnerShape<C>::MAXO, GL = 64 / PPW;
  constexpr int LPO = C >= 64 ? C / 8 : 1, ROWS = GL / LPO, CH = C >= 64 ? 8 : C;
  const int lane = (threadIdx.x & 63) % GL, row = lane / LPO, sub = lane % LPO;
  const int slot = (threadIdx.x >> 6) * PPW + (threadIdx.x & 63) / GL;   // this point's staging slot in LDS
  const int64_t p = (int64_t)blockIdx.x * 4 * PPW + slot;
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
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_inner_points_occ2(const InnerArgs a) {
  __shared__ double sh_obs[4 * InnerShape<C>::PPW][InnerShape<C>::MAXO][26];   // per point: q(4) t(3) k(12) sx sy corner(2) model patch
  inner_points_body<ST, C, FS>(a, sh_obs);
}
template <typename ST, int C, bool FS>
__global__ __launch_bounds__(256) void k_inner_points(const InnerArgs a) {
  __shared__ double sh_obs[4 * InnerShape<C>::PPW][InnerShape<C>::MAXO][26];
  inner_points_body<ST, C, FS>(a, sh_obs);
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
  const int ppb = arena->C >= 64 ? 4 : 32;   // points per workgroup (InnerShape)
  const unsigned blocks = (unsigned)((view->n_points + ppb - 1) / ppb);
  if (blocks == 0) return PXR_OK;
#define INNER_LAUNCH(KERNEL, ST, CC)                                                                          \
  do {                                                                                                        \
    if (cfg->use_float_simd) hipLaunchKernelGGL((KERNEL<ST, CC, true>), dim3(blocks), dim3(256), 0, ctx->stream, a);  \
    else hipLaunchKernelGGL((KERNEL<ST, CC, false>), dim3(blocks), dim3(256), 0, ctx->stream, a);             \
  } while (0)
  if (arena->C <= 4) {
    if (arena->dtype == PXR_F16 && arena->C == 3) INNER_LAUNCH(k_inner_points, _Float16, 3);
    else if (arena->dtype == PXR_F16) INNER_LAUNCH(k_inner_points, _Float16, 1);
    else if (arena->dtype == PXR_F32 && arena->C == 3) INNER_LAUNCH(k_inner_points, float, 3);
    else if (arena->dtype == PXR_F32) INNER_LAUNCH(k_inner_points, float, 1);
    else if (arena->C == 3) INNER_LAUNCH(k_inner_points, double, 3);
    else INNER_LAUNCH(k_inner_points, double, 1);
  } else if (arena->dtype == PXR_F16 && arena->C == 128) INNER_LAUNCH(k_inner_points_occ2, _Float16, 128);
  else if (arena->dtype == PXR_F16 && arena->C == 64) INNER_LAUNCH(k_inner_points_occ2, _Float16, 64);
  else if (arena->dtype == PXR_F32 && arena->C == 128) INNER_LAUNCH(k_inner_points_occ2, float, 128);
  else if (arena->dtype == PXR_F32 && arena->C == 64) INNER_LAUNCH(k_inner_points_occ2, float, 64);
  else if (arena->dtype == PXR_F64 && arena->C == 128) INNER_LAUNCH(k_inner_points, double, 128);
  else INNER_LAUNCH(k_inner_points, double, 64);
#undef INNER_LAUNCH
  return hip_check(hipGetLastError(), "k_inner_points launch");
}

}  // namespace pxr

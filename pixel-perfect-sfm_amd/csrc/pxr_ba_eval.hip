// pxr_ba_eval.hip -- the featuremetric BA residual kernel for gfx950 (MI355X).
//
// One unit of work = one observation = one residual block of C residuals:
//   (x,y) = WorldToPixel(camera, q, t, X)                         base/src/projection.h:60-75
//   f     = normalised bicubic descriptor of the obs' patch at (x,y)
//                                                                 base/src/interpolation.h:177-218,642-677
//   r     = f - ref_descriptor(point)                             residuals/src/feature_reference.h:132-134
// The reference materialises a C x (10+K) Jacobian per block through ceres::Jet
// (interpolation.h:130-140).  Because J = [gx gy] * d(x,y)/dparams, everything
// C-dimensional collapses to six dot products, which this kernel reduces inside the wave.
//
// Mapping (wave64, CDNA4): an observation occupies one DPP row = 16 lanes, each lane owns 8
// consecutive channels (16 B of fp16 per texel), so
//   * every texel fetch is one global_load_dwordx4 per lane, 256 B contiguous per row,
//     and the 4 x 4 stencil = 16 such loads in flight per lane (HBM-latency cover);
//   * the horizontal / vertical Catmull-Rom passes need no cross-lane traffic at all;
//   * the channel reductions are 4 DPP steps inside the row (no LDS, no ds_bpermute).
// A row processes 16 consecutive observations: lane s first computes the projection of
// observation s (so the fp64 geometry is not replicated 16x), then the row walks the 16
// observations, broadcasting (u,v,patch,point) from the owning lane.  At the end lane s holds
// the 64-byte record of observation s and the row stores 1 KiB contiguously.
//
// Precision contract = the reference's: fp32 FMA horizontal pass on fp16/fp32 texels, fp64
// vertical pass, fp64 normalisation (use_float_simd: fp32 vertical pass), identical
// operation order -> bit-identical h/f values; only the 16-lane reduction order differs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "pxr_device.h"
#include "pxr_interp.h"
#include "pxr_internal.h"

namespace pxr {

struct BaEvalArgs {
  pxr_ba_view v;
  const void* arena;
  const int32_t* corners;
  const double* scales;
  int H, W;
  double up;          // upsampling_factor_ of the patches (cost maps only; feature patches: 1)
  int l2_normalize;
  int check_bounds;
  pxr_loss loss;      // used when cost_out != NULL
  double* cost_out;   // NULL, or += sum over the launch of 0.5 rho(|r|^2) (fused cost reduction for the LM loop)
  double* rec;
  double* out_r;
  double* out_gx;
  double* out_gy;
  int opr;            // observations a lane group walks (<= LPO): 4 from 200k observations on, LPO below (launch_eval) -- same arithmetic
                      // per observation, only the number of wavefronts in flight changes
};

template <typename ST, int C, bool WITH_JAC, bool FLOAT_SIMD>
__global__ __launch_bounds__(256) void ba_eval_kernel(const BaEvalArgs a) {
  constexpr int LPO = C / 8;            // lanes per observation (16 for C = 128)
  constexpr int GPW = 64 / LPO;         // observation groups per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane & (LPO - 1);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int opr = a.opr;                                    // observations per lane group (<= LPO)
  const int64_t obs0 = (wave * GPW + (lane / LPO)) * opr;   // first observation of this lane group
  const int64_t n = a.v.n_obs;
  if (wave * GPW * opr >= n) return;                        // whole wave out of range (uniform)

  // ---- prologue: lane `sub` (< opr) owns the geometry of observation obs0 + sub -----------------
  const int64_t mine = obs0 + sub;
  const bool mine_valid = sub < opr && mine < n;
  const int64_t oi = mine_valid ? mine : n - 1;
  const int img = a.v.d_obs_image[oi];
  const int pt = a.v.d_obs_point[oi];
  const int64_t pidx = a.v.d_obs_patch[oi];
  const int cam = a.v.d_image_camera[img];
  double q[4], t[3], X[3], k[PXR_KPAD];
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = a.v.d_qvec[4 * (size_t)img + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) { t[i] = a.v.d_tvec[3 * (size_t)img + i]; X[i] = a.v.d_xyz[3 * (size_t)pt + i]; }
#pragma unroll
  for (int i = 0; i < PXR_KPAD; ++i) k[i] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + i];
  double my_x, my_y;
  world_to_pixel(a.v.d_cam_model[cam], k, q, t, X, my_x, my_y);
  // FeaturePatch::ToPixelCoordinates, featurepatch.h:250-255 (upsampling_factor_ = 1)
  const double sx = a.scales[2 * pidx], sy = a.scales[2 * pidx + 1];
  const double my_u = (my_x * sx - 0.5 - (double)a.corners[2 * pidx]);
  const double my_v = (my_y * sy - 0.5 - (double)a.corners[2 * pidx + 1]);

  double rec[PXR_OBS_REC];
#pragma unroll
  for (int i = 0; i < PXR_OBS_REC; ++i) rec[i] = 0.0;
  rec[6] = my_x; rec[7] = my_y;

  const size_t patch_elems = (size_t)a.H * a.W * C;
  const ST* arena = reinterpret_cast<const ST*>(a.arena);
  const int row_base = lane & ~(LPO - 1);

  for (int it = 0; it < opr; ++it) {
    if (obs0 + it >= n) break;   // uniform within the lane group
    const int src = row_base | it;
    const double u = shfl_f64(my_u, src), v = shfl_f64(my_v, src);
    const int64_t pi = shfl_i64(pidx, src);
    const int pti = __shfl(pt, src);
    // reference descriptor slice of this lane: 8 doubles (issued before the texel math)
    // (d_refs == NULL: no reference is subtracted -- the descriptor pass of the reference extraction)
    double2 rf0 = make_double2(0, 0), rf1 = rf0, rf2 = rf0, rf3 = rf0;
    if (a.v.d_refs) {
      const double* refp = a.v.d_refs + (size_t)pti * C + sub * 8;
      rf0 = *reinterpret_cast<const double2*>(refp);
      rf1 = *reinterpret_cast<const double2*>(refp + 2);
      rf2 = *reinterpret_cast<const double2*>(refp + 4);
      rf3 = *reinterpret_cast<const double2*>(refp + 6);
    }

    double f[8], fr[8], fc[8];
    interp8<ST, LPO, WITH_JAC, FLOAT_SIMD>(arena + (size_t)pi * patch_elems, a.H, a.W, C, sub, u, v,
                                           a.l2_normalize != 0, f, fr, fc);
    const double ref[8] = {rf0.x, rf0.y, rf1.x, rf1.y, rf2.x, rf2.y, rf3.x, rf3.y};
    double r[8];
    double s = 0, gcc = 0, gcr = 0, grr = 0, bc = 0, br = 0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      r[ch] = f[ch] - ref[ch];
      s = fma(r[ch], r[ch], s);
      if (WITH_JAC) {
        gcc = fma(fc[ch], fc[ch], gcc);
        gcr = fma(fc[ch], fr[ch], gcr);
        grr = fma(fr[ch], fr[ch], grr);
        bc = fma(fc[ch], r[ch], bc);
        br = fma(fr[ch], r[ch], br);
      }
    }
    if (LPO == 16) {
      s = row16_sum(s);
      if (WITH_JAC) { gcc = row16_sum(gcc); gcr = row16_sum(gcr); grr = row16_sum(grr); bc = row16_sum(bc); br = row16_sum(br); }
    } else {
      s = row8_sum(s);
      if (WITH_JAC) { gcc = row8_sum(gcc); gcr = row8_sum(gcr); grr = row8_sum(grr); bc = row8_sum(bc); br = row8_sum(br); }
    }
    if (sub == it) { rec[0] = s; rec[1] = gcc; rec[2] = gcr; rec[3] = grr; rec[4] = bc; rec[5] = br; }

    if (a.out_r) {   // materialise mode (parity checks): r, dr/dx, dr/dy per channel
      const double ax = shfl_f64(sx, src), ay = shfl_f64(sy, src);
      const size_t o = (size_t)(obs0 + it) * C + sub * 8;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        a.out_r[o + ch] = r[ch];
        if (WITH_JAC && a.out_gx) { a.out_gx[o + ch] = fc[ch] * ax; a.out_gy[o + ch] = fr[ch] * ay; }
      }
    }
  }
  if (mine_valid) {
    // Jet bridge: d/dx = dfdc * sx, d/dy = dfdr * sy  (interpolation.h:130-140 + featurepatch.h:250-255)
    rec[1] *= sx * sx; rec[2] *= sx * sy; rec[3] *= sy * sy; rec[4] *= sx; rec[5] *= sy;
    // InterpolationConfig.check_bounds (patch_interpolator.h:125-135,160-166): outside 0 < u < W, 0 < v < H the
    // interpolator reports `false` -- which FeatureReferenceCostFunctor passes on ONLY when it has no reference
    // descriptor (`if (!ref_descriptor_) return is_inside; ... return true;`, feature_reference.h:128-136: the cost-map
    // functor); with a reference the evaluation succeeds regardless.  A failed evaluation: the block's squared norm
    // becomes NaN, the cost non-finite -> the solver rejects the step (FAILURE at the initial point), like Ceres
    if (a.check_bounds && !a.v.d_refs && !(my_u > 0.0 && my_u < (double)a.W && my_v > 0.0 && my_v < (double)a.H)) rec[0] = __builtin_nan("");
    double2* o = reinterpret_cast<double2*>(a.rec + (size_t)mine * PXR_OBS_REC);
    o[0] = make_double2(rec[0], rec[1]);
    o[1] = make_double2(rec[2], rec[3]);
    o[2] = make_double2(rec[4], rec[5]);
    o[3] = make_double2(rec[6], rec[7]);
  }
  if (a.cost_out) {   // one atomic per wavefront; saves the solver a separate pass over the records
    double c = 0.0;
    if (mine_valid) {
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, rec[0], rho);
      c = 0.5 * rho[0];
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.cost_out, c);
  }
}

// ---- few-channel patches: cost-map BA ------------------------------------------------------------
// CostMapBundleOptimizer (bundle_adjustment/src/costmap_bundle_optimizer.h:76-132) runs the same functor on
// 1- or 3-channel cost maps with no reference descriptor (nullptr, "just minimize"): the residual block IS the
// interpolated texel.  16 texels of 2..24 bytes per observation: one lane per observation, the record and the
// fused cost exactly as above, so the whole Schur pipeline is shared.
template <typename ST, int C, bool WITH_JAC>
__global__ __launch_bounds__(256) void ba_eval_small_kernel(const BaEvalArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = a.v.n_obs;
  const bool valid = i < n;
  double rec[PXR_OBS_REC];
#pragma unroll
  for (int j = 0; j < PXR_OBS_REC; ++j) rec[j] = 0.0;
  if (valid) {
    const int img = a.v.d_obs_image[i];
    const int pt = a.v.d_obs_point[i];
    const int64_t pidx = a.v.d_obs_patch[i];
    const int cam = a.v.d_image_camera[img];
    double q[4], t[3], X[3], k[PXR_KPAD];
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = a.v.d_qvec[4 * (size_t)img + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) { t[j] = a.v.d_tvec[3 * (size_t)img + j]; X[j] = a.v.d_xyz[3 * (size_t)pt + j]; }
#pragma unroll
    for (int j = 0; j < PXR_KPAD; ++j) k[j] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + j];
    double x, y;
    world_to_pixel(a.v.d_cam_model[cam], k, q, t, X, x, y);
    const double sx = a.scales[2 * pidx], sy = a.scales[2 * pidx + 1];
    // FeaturePatch::ToPixelCoordinates (featurepatch.h:250-255) incl. the upsampling factor of cost maps
    const double u = (x * sx - 0.5 - (double)a.corners[2 * pidx]) * a.up;
    const double v = (y * sy - 0.5 - (double)a.corners[2 * pidx + 1]) * a.up;
    double f[C], fr[C], fc[C];
    interp_small<ST, C>(reinterpret_cast<const ST*>(a.arena) + (size_t)pidx * a.H * a.W * C, a.H, a.W, u, v,
                        a.l2_normalize != 0, f, fr, fc);
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      const double r = a.v.d_refs ? f[ch] - a.v.d_refs[(size_t)pt * C + ch] : f[ch];
      rec[0] = fma(r, r, rec[0]);
      if (WITH_JAC) {
        rec[1] = fma(fc[ch], fc[ch], rec[1]); rec[2] = fma(fc[ch], fr[ch], rec[2]); rec[3] = fma(fr[ch], fr[ch], rec[3]);
        rec[4] = fma(fc[ch], r, rec[4]); rec[5] = fma(fr[ch], r, rec[5]);
      }
      if (a.out_r) {
        a.out_r[(size_t)i * C + ch] = r;
        if (WITH_JAC && a.out_gx) { a.out_gx[(size_t)i * C + ch] = fc[ch] * sx * a.up; a.out_gy[(size_t)i * C + ch] = fr[ch] * sy * a.up; }
      }
    }
    { const double ux = sx * a.up, uy = sy * a.up; rec[1] *= ux * ux; rec[2] *= ux * uy; rec[3] *= uy * uy; rec[4] *= ux; rec[5] *= uy; }
    rec[6] = x; rec[7] = y;
    if (a.check_bounds && !a.v.d_refs && !(u > 0.0 && u < (double)a.W && v > 0.0 && v < (double)a.H)) rec[0] = __builtin_nan("");   // no reference: the cost-map functor
    double2* o = reinterpret_cast<double2*>(a.rec + (size_t)i * PXR_OBS_REC);
    o[0] = make_double2(rec[0], rec[1]);
    o[1] = make_double2(rec[2], rec[3]);
    o[2] = make_double2(rec[4], rec[5]);
    o[3] = make_double2(rec[6], rec[7]);
  }
  if (a.cost_out) {
    double c = 0.0;
    if (valid) {
      double rho[3];
      loss_eval(a.loss.type, a.loss.a, 1.0, rec[0], rho);
      c = 0.5 * rho[0];
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.cost_out, c);
  }
}

template <typename ST, int C>
static int launch_eval_small(pxr_ctx* ctx, const BaEvalArgs& a, bool with_jac) {
  const int64_t blocks = (a.v.n_obs + 255) / 256;
  if (blocks == 0) return PXR_OK;
  if (with_jac) hipLaunchKernelGGL((ba_eval_small_kernel<ST, C, true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL((ba_eval_small_kernel<ST, C, false>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
  return pxr::hip_check(hipGetLastError(), "ba_eval_small_kernel launch");
}

// ---- projection Jacobians (parity checks) ------------------------------------------------
__global__ __launch_bounds__(256) void ba_projjac_kernel(const pxr_ba_view v, double* __restrict__ P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v.n_obs) return;
  const int img = v.d_obs_image[i], pt = v.d_obs_point[i], cam = v.d_image_camera[img];
  double q[4], t[3], X[3], k[PXR_KPAD];
  for (int j = 0; j < 4; ++j) q[j] = v.d_qvec[4 * (size_t)img + j];
  for (int j = 0; j < 3; ++j) { t[j] = v.d_tvec[3 * (size_t)img + j]; X[j] = v.d_xyz[3 * (size_t)pt + j]; }
  for (int j = 0; j < PXR_KPAD; ++j) k[j] = v.d_cam_params[(size_t)cam * PXR_KPAD + j];
  double x, y, A[2][3], Pq[2][4], PX[2][3], Pk[2][PXR_KPAD];
  world_to_pixel_jac(v.d_cam_model[cam], k, q, t, X, x, y, A, Pq, PX, Pk);
  const int NJ = 10 + PXR_KPAD;
  double* o = P + (size_t)i * 2 * NJ;
  for (int r = 0; r < 2; ++r) {
    for (int j = 0; j < 4; ++j) o[r * NJ + j] = Pq[r][j];
    for (int j = 0; j < 3; ++j) o[r * NJ + 4 + j] = A[r][j];   // dp/dt = I
    for (int j = 0; j < 3; ++j) o[r * NJ + 7 + j] = PX[r][j];
    for (int j = 0; j < PXR_KPAD; ++j) o[r * NJ + 10 + j] = Pk[r][j];
  }
}

// ---- cost = sum 0.5 rho(s) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void ba_cost_kernel(const double* __restrict__ rec, int64_t n,
                                                      pxr_loss loss, double* __restrict__ out) {
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double rho[3];
    loss_eval(loss.type, loss.a, 1.0, rec[i * PXR_OBS_REC], rho);
    acc += 0.5 * rho[0];
  }
  // wave reduce then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) acc += shfl_f64(acc, (threadIdx.x & 63) ^ off);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

template <typename ST, int C>
static int launch_eval(pxr_ctx* ctx, const BaEvalArgs& a_in, bool with_jac, bool float_simd) {
  constexpr int LPO = C / 8;
  BaEvalArgs a = a_in;
  // How many observations a lane group walks (round 6; until then always LPO = 16: lane s computes the projection of observation
  // s, then the row walks its 16 observations).  With 4 the launch has four times the wavefronts -- each with sixteen 16-byte
  // loads per lane in flight -- and a quarter of the projection lanes idle: measured on MI355X, alternating runs of the bench
  // (ms per step, 16 / 4 observations per group): 1M observations 0.802-0.813 / 0.768-0.793, 500k 0.404 / 0.388, 250k 0.206 /
  // 0.202, 125k 0.1046 / 0.1066 (6 and 8 per group: worse than both below 250k).  PXR_BA_EVAL_OPR=<n> forces a value.
  static const int opr_knob = std::getenv("PXR_BA_EVAL_OPR") ? atoi(std::getenv("PXR_BA_EVAL_OPR")) : 0;
  a.opr = opr_knob > 0 ? std::min(opr_knob, LPO) : (a.v.n_obs >= 200000 ? std::min(4, LPO) : LPO);
  const int64_t obs_per_block = 4 * (64 / LPO) * a.opr;   // 4 waves x (64 / LPO groups x opr observations)
  const int64_t blocks = (a.v.n_obs + obs_per_block - 1) / obs_per_block;
  if (blocks == 0) return PXR_OK;
  dim3 grid((unsigned)blocks), block(256);
  if (with_jac) {
    if (float_simd) hipLaunchKernelGGL((ba_eval_kernel<ST, C, true, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((ba_eval_kernel<ST, C, true, false>), grid, block, 0, ctx->stream, a);
  } else {
    if (float_simd) hipLaunchKernelGGL((ba_eval_kernel<ST, C, false, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((ba_eval_kernel<ST, C, false, false>), grid, block, 0, ctx->stream, a);
  }
  return pxr::hip_check(hipGetLastError(), "ba_eval_kernel launch");
}

template <typename ST>
static int launch_eval_c(pxr_ctx* ctx, int C, const BaEvalArgs& a, bool with_jac, bool float_simd) {
  switch (C) {
    case 128: return launch_eval<ST, 128>(ctx, a, with_jac, float_simd);
    case 64: return launch_eval<ST, 64>(ctx, a, with_jac, float_simd);
    case 3: return launch_eval_small<ST, 3>(ctx, a, with_jac);   // cost maps (costmap_bundle_optimizer.h:9-14)
    case 1: return launch_eval_small<ST, 1>(ctx, a, with_jac);
    default:
      return set_error(PXR_EUNSUPPORTED, "pxr_ba_eval: CHANNELS=%d not supported (128, 64; cost maps: 3, 1)", C);
  }
}

}  // namespace pxr

extern "C" {

int pxr_ba_eval(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                int with_jacobian, double* d_rec, double* d_r, double* d_gx, double* d_gy) {
  return pxr::ba_eval_with_cost(ctx, arena, view, cfg, with_jacobian, d_rec, d_r, d_gx, d_gy, nullptr, nullptr);
}

}  // extern "C"

int pxr::ba_eval_with_cost(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                           int with_jacobian, double* d_rec, double* d_r, double* d_gx, double* d_gy,
                           const pxr_loss* loss, double* d_cost_sum) {
  PXR_REQUIRE(ctx && arena && view && cfg && d_rec, "pxr_ba_eval: NULL argument");
  PXR_REQUIRE(!d_cost_sum || loss, "pxr_ba_eval: the fused cost needs a loss");
  PXR_REQUIRE(view->n_obs >= 0, "pxr_ba_eval: negative n_obs");
  PXR_REQUIRE(arena->up == 1.0 || arena->C <= 4, "pxr_ba_eval: an upsampling factor is a property of cost maps (1 / 3 channels)");
  PXR_REQUIRE((d_gx == nullptr) == (d_gy == nullptr), "pxr_ba_eval: d_gx and d_gy must be given together");
  PXR_REQUIRE(!(d_gx && !d_r), "pxr_ba_eval: d_gx/d_gy require d_r");
  if (view->n_obs == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
  pxr::BaEvalArgs a;
  a.v = *view;
  a.arena = arena->d_data; a.corners = arena->d_corners; a.scales = arena->d_scales;
  a.H = arena->H; a.W = arena->W; a.up = arena->up; a.l2_normalize = cfg->l2_normalize; a.check_bounds = cfg->check_bounds;
  a.rec = d_rec; a.out_r = d_r; a.out_gx = d_gx; a.out_gy = d_gy;
  a.cost_out = d_cost_sum;
  if (loss) a.loss = *loss; else { a.loss.type = PXR_LOSS_TRIVIAL; a.loss.a = 1.0; }
  const bool wj = with_jacobian != 0, fs = cfg->use_float_simd != 0;
  switch (arena->dtype) {
    case PXR_F16: return pxr::launch_eval_c<_Float16>(ctx, arena->C, a, wj, fs);
    case PXR_F32: return pxr::launch_eval_c<float>(ctx, arena->C, a, wj, fs);
    case PXR_F64: return pxr::launch_eval_c<double>(ctx, arena->C, a, wj, fs);
  }
  return pxr::set_error(PXR_EINVAL, "pxr_ba_eval: bad arena dtype");
}

extern "C" {

int pxr_ba_projection_jacobian(pxr_ctx* ctx, const pxr_ba_view* view, double* d_P) {
  PXR_REQUIRE(ctx && view && d_P, "pxr_ba_projection_jacobian: NULL argument");
  if (view->n_obs == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
  const unsigned blocks = (unsigned)((view->n_obs + 255) / 256);
  hipLaunchKernelGGL(pxr::ba_projjac_kernel, dim3(blocks), dim3(256), 0, ctx->stream, *view, d_P);
  return pxr::hip_check(hipGetLastError(), "ba_projjac_kernel launch");
}

int pxr_ba_cost(pxr_ctx* ctx, const double* d_rec, int64_t n_obs, const pxr_loss* loss, double* h_cost) {
  PXR_REQUIRE(ctx && d_rec && loss && h_cost, "pxr_ba_cost: NULL argument");
  PXR_HIP(hipSetDevice(ctx->device));
  PXR_HIP(hipMemsetAsync(ctx->d_scratch, 0, sizeof(double), ctx->stream));
  if (n_obs > 0) {
    int64_t blocks = (n_obs + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pxr::ba_cost_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_rec, n_obs,
                       *loss, ctx->d_scratch);
    PXR_HIP(hipGetLastError());
  }
  PXR_HIP(hipMemcpyAsync(h_cost, ctx->d_scratch, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  PXR_HIP(hipStreamSynchronize(ctx->stream));
  return PXR_OK;
}

}  // extern "C"

// pxr_refs.hip -- reference-descriptor extraction for featuremetric BA on gfx950.
//
// Replaces ReferenceExtractor::Run / RunSubset / ComputeReference / FillDescriptorTrack
// (bundle_adjustment/src/reference_extractor.h:125-318) and RobustMeanIRLS
// (base/src/irls_optim.h:24-71): per 3D point, interpolate the normalised descriptor of every
// observation at the point's CURRENT projection (value only), run the IRLS robust mean
// (weights 1/rho(|d_i - mu|^2) with the loss VALUE, early return if rho <= 0, 100 iterations),
// and keep the observation descriptor closest to the robust mean (closest_to_robust_mean = true,
// reference_extractor.h:63,249-272).
//
// Pass 1 reuses the fused BA kernel with no reference (d_refs = NULL) in materialise mode: the
// descriptors land in an [n_obs][C] fp64 scratch.  Pass 2: one 16-lane row per point, a lane owns
// 8 channels; tracks of up to 8 observations keep all descriptors in registers for the 100 IRLS
// iterations (no memory traffic in the loop), longer tracks re-read them from L2.
#include <hip/hip_runtime.h>

#include <vector>

#include "pxr_device.h"
#include "pxr_interp.h"
#include "pxr_internal.h"

namespace pxr {

constexpr int REF_REG_TRACK = 8;

template <int C>
__global__ __launch_bounds__(256) void k_irls(int64_t n_points, const int64_t* __restrict__ pt_ptr,
                                              const int64_t* __restrict__ pt_obs,
                                              const double* __restrict__ desc, pxr_loss loss, int iters,
                                              int l2_normalize, double* __restrict__ wbuf,
                                              double* __restrict__ refs, int64_t* __restrict__ ref_obs,
                                              double* __restrict__ robust_mean) {
  constexpr int LPO = C / 8, G = 256 / LPO;
  const int sub = threadIdx.x % LPO;
  const int64_t p = (int64_t)blockIdx.x * G + threadIdx.x / LPO;
  if (p >= n_points) return;
  const int64_t o0 = pt_ptr[p], o1 = pt_ptr[p + 1];
  const int n = (int)(o1 - o0);
  if (n == 0) { if (sub == 0) ref_obs[p] = -1; return; }
  auto rsum = [](double v) { return LPO == 16 ? row16_sum(v) : row8_sum(v); };

  double mu[8];
  int early = -1;
  if (n <= REF_REG_TRACK) {
    double d[REF_REG_TRACK][8], w[REF_REG_TRACK];
#pragma unroll
    for (int i = 0; i < REF_REG_TRACK; ++i) {
      w[i] = 1.0;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) d[i][ch] = (i < n) ? desc[(size_t)pt_obs[o0 + i] * C + sub * 8 + ch] : 0.0;
    }
    for (int k = 0; k < iters && early < 0; ++k) {
      double sw = 0.0;
#pragma unroll
      for (int i = 0; i < REF_REG_TRACK; ++i) if (i < n) sw += w[i];
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) mu[ch] = 0.0;
#pragma unroll
      for (int i = 0; i < REF_REG_TRACK; ++i) {
        if (i >= n) continue;
        w[i] = w[i] / sw;                                    // irls_optim.h:44
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) mu[ch] += d[i][ch] * w[i];   // :46-48
      }
      if (l2_normalize) {                                    // :54-58
        double ss = 0.0;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) ss = fma(mu[ch], mu[ch], ss);
        const double nrm = sqrt(rsum(ss));
        if (nrm > 0.0) {
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) mu[ch] /= nrm;
        }
      }
#pragma unroll
      for (int i = 0; i < REF_REG_TRACK; ++i) {              // :60-69
        if (i >= n || early >= 0) continue;
        double s = 0.0;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) { const double df = d[i][ch] - mu[ch]; s = fma(df, df, s); }
        s = rsum(s);
        double rho[3];
        loss_eval(loss.type, loss.a, 1.0, s, rho);
        if (rho[0] > 0.0) w[i] = 1.0 / rho[0];
        else {
          early = i;
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) mu[ch] = d[i][ch];
        }
      }
    }
  } else {
    double* w = wbuf + o0;   // per-observation weights in L2-resident scratch (all lanes of the row write the same value)
    for (int i = 0; i < n; ++i) w[i] = 1.0;
    for (int k = 0; k < iters && early < 0; ++k) {
      double sw = 0.0;
      for (int i = 0; i < n; ++i) sw += w[i];
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) mu[ch] = 0.0;
      for (int i = 0; i < n; ++i) {
        const double wi = w[i] / sw;
        w[i] = wi;
        const double* di = desc + (size_t)pt_obs[o0 + i] * C + sub * 8;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) mu[ch] += di[ch] * wi;
      }
      if (l2_normalize) {
        double ss = 0.0;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) ss = fma(mu[ch], mu[ch], ss);
        const double nrm = sqrt(rsum(ss));
        if (nrm > 0.0) {
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) mu[ch] /= nrm;
        }
      }
      for (int i = 0; i < n && early < 0; ++i) {
        const double* di = desc + (size_t)pt_obs[o0 + i] * C + sub * 8;
        double s = 0.0;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) { const double df = di[ch] - mu[ch]; s = fma(df, df, s); }
        s = rsum(s);
        double rho[3];
        loss_eval(loss.type, loss.a, 1.0, s, rho);
        if (rho[0] > 0.0) w[i] = 1.0 / rho[0];
        else {
          early = i;
#pragma unroll
          for (int ch = 0; ch < 8; ++ch) mu[ch] = di[ch];
        }
      }
    }
  }
  // ComputeReference: observation closest to the robust mean (first minimum, Eigen minCoeff)
  int best = 0;
  double bestd = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* di = desc + (size_t)pt_obs[o0 + i] * C + sub * 8;
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { const double df = di[ch] - mu[ch]; s = fma(df, df, s); }
    s = rsum(s);
    if (i == 0 || s < bestd) { bestd = s; best = i; }
  }
  const int64_t bo = pt_obs[o0 + best];
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    refs[(size_t)p * C + sub * 8 + ch] = desc[(size_t)bo * C + sub * 8 + ch];
    if (robust_mean) robust_mean[(size_t)p * C + sub * 8 + ch] = mu[ch];
  }
  if (sub == 0) ref_obs[p] = bo;
}

// ---- FindNearestReferences (localization/src/nearest_references.h:20-52) ---------------------------------
// One DPP row (C/8 lanes) per 2D-3D correspondence: interpolate the query descriptor once, then scan
// the candidate descriptors (the observations of its 3D point) for the smallest squared distance;
// the first minimum wins, like the reference's strict `d < min_distance`.
template <typename ST, int C>
__global__ __launch_bounds__(256) void k_nearest(int64_t n, const ST* __restrict__ arena, const int32_t* __restrict__ corners,
                                                 const double* __restrict__ scales, int H, int W, int l2_normalize,
                                                 int float_simd, const double* __restrict__ kp,
                                                 const int64_t* __restrict__ patch, const int64_t* __restrict__ cand_ptr,
                                                 const int64_t* __restrict__ cand_index, const double* __restrict__ cand_desc,
                                                 int64_t* __restrict__ best, double* __restrict__ best_dist,
                                                 double* __restrict__ out_desc) {
  constexpr int LPO = C / 8, G = 256 / LPO;
  const int sub = threadIdx.x % LPO;
  const int64_t i = (int64_t)blockIdx.x * G + threadIdx.x / LPO;
  if (i >= n) return;
  const int64_t pi = patch[i];
  const double sx = scales[2 * pi], sy = scales[2 * pi + 1];
  const double u = kp[2 * i] * sx - 0.5 - (double)corners[2 * pi];          // featurepatch.h:250-255
  const double v = kp[2 * i + 1] * sy - 0.5 - (double)corners[2 * pi + 1];
  const ST* p = arena + (size_t)pi * H * W * C;
  double f[8], fr[8], fc[8];
  if (float_simd) interp8<ST, LPO, false, true>(p, H, W, C, sub, u, v, l2_normalize != 0, f, fr, fc);
  else interp8<ST, LPO, false, false>(p, H, W, C, sub, u, v, l2_normalize != 0, f, fr, fc);
  double dmin = 1.7976931348623157e308;
  int64_t bi = -1;
  for (int64_t o = cand_ptr[i]; o < cand_ptr[i + 1]; ++o) {
    const int64_t row = cand_index ? cand_index[o] : o;
    const double* d = cand_desc + (size_t)row * C + sub * 8;
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { const double e = d[ch] - f[ch]; s = fma(e, e, s); }
    s = LPO == 16 ? row16_sum(s) : row8_sum(s);
    if (s < dmin) { dmin = s; bi = row; }
  }
  if (sub == 0) { best[i] = bi; if (best_dist) best_dist[i] = dmin; }
  if (out_desc && bi >= 0) {
    const double* d = cand_desc + (size_t)bi * C + sub * 8;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) out_desc[(size_t)i * C + sub * 8 + ch] = d[ch];
  }
}

// ---- PatchInterpolator::Evaluate (features/src/patch_interpolator.h:86-135), batched -------------------------
template <typename ST, int C>
__global__ __launch_bounds__(256) void k_interpolate(int64_t n, const ST* __restrict__ arena, const int32_t* __restrict__ corners,
                                                     const double* __restrict__ scales, int H, int W, int l2_normalize,
                                                     int float_simd, const double* __restrict__ kp,
                                                     const int64_t* __restrict__ patch, double* __restrict__ out_f,
                                                     double* __restrict__ out_J) {
  constexpr int LPO = C / 8, G = 256 / LPO;
  const int sub = threadIdx.x % LPO;
  const int64_t i = (int64_t)blockIdx.x * G + threadIdx.x / LPO;
  if (i >= n) return;
  const int64_t pi = patch[i];
  const double sx = scales[2 * pi], sy = scales[2 * pi + 1];
  const double u = kp[2 * i] * sx - 0.5 - (double)corners[2 * pi];
  const double v = kp[2 * i + 1] * sy - 0.5 - (double)corners[2 * pi + 1];
  const ST* p = arena + (size_t)pi * H * W * C;
  double f[8], fr[8], fc[8];
  if (out_J) {
    if (float_simd) interp8<ST, LPO, true, true>(p, H, W, C, sub, u, v, l2_normalize != 0, f, fr, fc);
    else interp8<ST, LPO, true, false>(p, H, W, C, sub, u, v, l2_normalize != 0, f, fr, fc);
  } else {
    if (float_simd) interp8<ST, LPO, false, true>(p, H, W, C, sub, u, v, l2_normalize != 0, f, fr, fc);
    else interp8<ST, LPO, false, false>(p, H, W, C, sub, u, v, l2_normalize != 0, f, fr, fc);
  }
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) {
    const size_t o = (size_t)i * C + sub * 8 + ch;
    out_f[o] = f[ch];
    if (out_J) { out_J[2 * o] = fc[ch] * sx; out_J[2 * o + 1] = fr[ch] * sy; }   // d/dx, d/dy in image coordinates
  }
}

// ---- few-channel patches (C = 3 or 1: image intensities; FeatureReferenceBundleOptimizer registers (3, 1) and (1, 1),
// feature_reference_bundle_optimizer.h:13-16) ---------------------------------------------------------------------------
// Below 8 channels the reference interpolates with the scalar [upstream] ceres::BiCubicInterpolator (interp_small); one lane
// holds the whole descriptor, so a point / correspondence / keypoint is ONE lane here and nothing crosses lanes.
template <int C>
__global__ __launch_bounds__(256) void k_irls_small(int64_t n_points, const int64_t* __restrict__ pt_ptr,
                                                    const int64_t* __restrict__ pt_obs, const double* __restrict__ desc,
                                                    pxr_loss loss, int iters, int l2_normalize, double* __restrict__ wbuf,
                                                    double* __restrict__ refs, int64_t* __restrict__ ref_obs,
                                                    double* __restrict__ robust_mean) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  const int64_t o0 = pt_ptr[p];
  const int n = (int)(pt_ptr[p + 1] - o0);
  if (n == 0) { ref_obs[p] = -1; return; }
  double* w = wbuf + o0;
  for (int i = 0; i < n; ++i) w[i] = 1.0;
  double mu[C];
  int early = -1;
  for (int k = 0; k < iters && early < 0; ++k) {
    double sw = 0.0;
    for (int i = 0; i < n; ++i) sw += w[i];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) mu[ch] = 0.0;
    for (int i = 0; i < n; ++i) {
      const double wi = w[i] / sw;                           // irls_optim.h:44
      w[i] = wi;
      const double* di = desc + (size_t)pt_obs[o0 + i] * C;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) mu[ch] += di[ch] * wi;  // :46-48
    }
    if (l2_normalize) {                                      // :54-58
      double ss = 0.0;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) ss = fma(mu[ch], mu[ch], ss);
      const double nrm = sqrt(ss);
      if (nrm > 0.0) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mu[ch] /= nrm;
      }
    }
    for (int i = 0; i < n && early < 0; ++i) {               // :60-69
      const double* di = desc + (size_t)pt_obs[o0 + i] * C;
      double s = 0.0;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) { const double df = di[ch] - mu[ch]; s = fma(df, df, s); }
      double rho[3];
      loss_eval(loss.type, loss.a, 1.0, s, rho);
      if (rho[0] > 0.0) w[i] = 1.0 / rho[0];
      else {
        early = i;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) mu[ch] = di[ch];
      }
    }
  }
  int best = 0;                                              // ComputeReference: first minimum, like Eigen's minCoeff
  double bestd = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* di = desc + (size_t)pt_obs[o0 + i] * C;
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) { const double df = di[ch] - mu[ch]; s = fma(df, df, s); }
    if (i == 0 || s < bestd) { bestd = s; best = i; }
  }
  const int64_t bo = pt_obs[o0 + best];
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    refs[(size_t)p * C + ch] = desc[(size_t)bo * C + ch];
    if (robust_mean) robust_mean[(size_t)p * C + ch] = mu[ch];
  }
  ref_obs[p] = bo;
}

template <typename ST, int C>
__global__ __launch_bounds__(256) void k_nearest_small(int64_t n, const ST* __restrict__ arena, const int32_t* __restrict__ corners,
                                                       const double* __restrict__ scales, int H, int W, int l2_normalize,
                                                       const double* __restrict__ kp, const int64_t* __restrict__ patch,
                                                       const int64_t* __restrict__ cand_ptr, const int64_t* __restrict__ cand_index,
                                                       const double* __restrict__ cand_desc, int64_t* __restrict__ best,
                                                       double* __restrict__ best_dist, double* __restrict__ out_desc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t pi = patch[i];
  const double u = kp[2 * i] * scales[2 * pi] - 0.5 - (double)corners[2 * pi];
  const double v = kp[2 * i + 1] * scales[2 * pi + 1] - 0.5 - (double)corners[2 * pi + 1];
  double f[C], fr[C], fc[C];
  interp_small<ST, C>(arena + (size_t)pi * H * W * C, H, W, u, v, l2_normalize != 0, f, fr, fc);
  double dmin = 1.7976931348623157e308;
  int64_t bi = -1;
  for (int64_t o = cand_ptr[i]; o < cand_ptr[i + 1]; ++o) {
    const int64_t row = cand_index ? cand_index[o] : o;
    const double* d = cand_desc + (size_t)row * C;
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) { const double e = d[ch] - f[ch]; s = fma(e, e, s); }
    if (s < dmin) { dmin = s; bi = row; }
  }
  best[i] = bi;
  if (best_dist) best_dist[i] = dmin;
  if (out_desc && bi >= 0) {
#pragma unroll
    for (int ch = 0; ch < C; ++ch) out_desc[(size_t)i * C + ch] = cand_desc[(size_t)bi * C + ch];
  }
}

template <typename ST, int C>
__global__ __launch_bounds__(256) void k_interpolate_small(int64_t n, const ST* __restrict__ arena, const int32_t* __restrict__ corners,
                                                           const double* __restrict__ scales, int H, int W, int l2_normalize,
                                                           const double* __restrict__ kp, const int64_t* __restrict__ patch,
                                                           double* __restrict__ out_f, double* __restrict__ out_J) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t pi = patch[i];
  const double sx = scales[2 * pi], sy = scales[2 * pi + 1];
  const double u = kp[2 * i] * sx - 0.5 - (double)corners[2 * pi];
  const double v = kp[2 * i + 1] * sy - 0.5 - (double)corners[2 * pi + 1];
  double f[C], fr[C], fc[C];
  interp_small<ST, C>(arena + (size_t)pi * H * W * C, H, W, u, v, l2_normalize != 0, f, fr, fc);
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    const size_t o = (size_t)i * C + ch;
    out_f[o] = f[ch];
    if (out_J) { out_J[2 * o] = fc[ch] * sx; out_J[2 * o + 1] = fr[ch] * sy; }
  }
}

}  // namespace pxr

extern "C" int pxr_interpolate(pxr_ctx* ctx, pxr_arena* arena, const pxr_interp_cfg* cfg, int64_t n, const double* d_kp,
                               const int64_t* d_patch, double* d_desc, double* d_J) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && cfg && d_kp && d_patch && d_desc, "pxr_interpolate: NULL argument");
  if (n == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
#define INTERP_LAUNCH(ST, CC)                                                                                      \
  hipLaunchKernelGGL((k_interpolate<ST, CC>), dim3((unsigned)((n + (256 / (CC / 8)) - 1) / (256 / (CC / 8)))), dim3(256), 0, \
                     ctx->stream, n, (const ST*)arena->d_data, arena->d_corners, arena->d_scales, arena->H, arena->W, \
                     cfg->l2_normalize, cfg->use_float_simd, d_kp, d_patch, d_desc, d_J)
  if (arena->dtype == PXR_F16 && arena->C == 128) INTERP_LAUNCH(_Float16, 128);
  else if (arena->dtype == PXR_F16 && arena->C == 64) INTERP_LAUNCH(_Float16, 64);
  else if (arena->dtype == PXR_F32 && arena->C == 128) INTERP_LAUNCH(float, 128);
  else if (arena->dtype == PXR_F32 && arena->C == 64) INTERP_LAUNCH(float, 64);
  else if (arena->dtype == PXR_F64 && arena->C == 128) INTERP_LAUNCH(double, 128);
  else if (arena->dtype == PXR_F64 && arena->C == 64) INTERP_LAUNCH(double, 64);
#define INTERP_SMALL(ST, CC)                                                                                       \
  hipLaunchKernelGGL((k_interpolate_small<ST, CC>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,   \
                     (const ST*)arena->d_data, arena->d_corners, arena->d_scales, arena->H, arena->W, cfg->l2_normalize, \
                     d_kp, d_patch, d_desc, d_J)
  else if (arena->dtype == PXR_F16 && arena->C == 3) INTERP_SMALL(_Float16, 3);
  else if (arena->dtype == PXR_F16 && arena->C == 1) INTERP_SMALL(_Float16, 1);
  else if (arena->dtype == PXR_F32 && arena->C == 3) INTERP_SMALL(float, 3);
  else if (arena->dtype == PXR_F32 && arena->C == 1) INTERP_SMALL(float, 1);
  else if (arena->dtype == PXR_F64 && arena->C == 3) INTERP_SMALL(double, 3);
  else if (arena->dtype == PXR_F64 && arena->C == 1) INTERP_SMALL(double, 1);
  else return set_error(PXR_EUNSUPPORTED, "pxr_interpolate: CHANNELS=%d not supported (128, 64, 3, 1)", arena->C);
#undef INTERP_SMALL
#undef INTERP_LAUNCH
  return hip_check(hipGetLastError(), "k_interpolate launch");
}

extern "C" int pxr_nearest_references(pxr_ctx* ctx, pxr_arena* arena, const pxr_interp_cfg* cfg, int64_t n,
                                      const double* d_kp, const int64_t* d_patch, const int64_t* d_cand_ptr,
                                      const int64_t* d_cand_index, const double* d_cand_desc, int64_t* d_best,
                                      double* d_best_dist, double* d_out_desc) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && cfg && d_kp && d_patch && d_cand_ptr && d_cand_desc && d_best, "pxr_nearest_references: NULL argument");
  if (n == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
#define NEAREST_LAUNCH(ST, CC)                                                                                     \
  hipLaunchKernelGGL((k_nearest<ST, CC>), dim3((unsigned)((n + (256 / (CC / 8)) - 1) / (256 / (CC / 8)))), dim3(256), 0,     \
                     ctx->stream, n, (const ST*)arena->d_data, arena->d_corners, arena->d_scales, arena->H, arena->W, \
                     cfg->l2_normalize, cfg->use_float_simd, d_kp, d_patch, d_cand_ptr, d_cand_index, d_cand_desc, d_best, \
                     d_best_dist, d_out_desc)
  if (arena->dtype == PXR_F16 && arena->C == 128) NEAREST_LAUNCH(_Float16, 128);
  else if (arena->dtype == PXR_F16 && arena->C == 64) NEAREST_LAUNCH(_Float16, 64);
  else if (arena->dtype == PXR_F32 && arena->C == 128) NEAREST_LAUNCH(float, 128);
  else if (arena->dtype == PXR_F32 && arena->C == 64) NEAREST_LAUNCH(float, 64);
  else if (arena->dtype == PXR_F64 && arena->C == 128) NEAREST_LAUNCH(double, 128);
  else if (arena->dtype == PXR_F64 && arena->C == 64) NEAREST_LAUNCH(double, 64);
#define NEAREST_SMALL(ST, CC)                                                                                      \
  hipLaunchKernelGGL((k_nearest_small<ST, CC>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, n,       \
                     (const ST*)arena->d_data, arena->d_corners, arena->d_scales, arena->H, arena->W, cfg->l2_normalize, \
                     d_kp, d_patch, d_cand_ptr, d_cand_index, d_cand_desc, d_best, d_best_dist, d_out_desc)
  else if (arena->dtype == PXR_F16 && arena->C == 3) NEAREST_SMALL(_Float16, 3);
  else if (arena->dtype == PXR_F16 && arena->C == 1) NEAREST_SMALL(_Float16, 1);
  else if (arena->dtype == PXR_F32 && arena->C == 3) NEAREST_SMALL(float, 3);
  else if (arena->dtype == PXR_F32 && arena->C == 1) NEAREST_SMALL(float, 1);
  else if (arena->dtype == PXR_F64 && arena->C == 3) NEAREST_SMALL(double, 3);
  else if (arena->dtype == PXR_F64 && arena->C == 1) NEAREST_SMALL(double, 1);
  else return set_error(PXR_EUNSUPPORTED, "pxr_nearest_references: CHANNELS=%d not supported (128, 64, 3, 1)", arena->C);
#undef NEAREST_SMALL
#undef NEAREST_LAUNCH
  return hip_check(hipGetLastError(), "k_nearest launch");
}

extern "C" int pxr_ba_compute_references(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view,
                                         const pxr_interp_cfg* cfg, const pxr_loss* loss, int iters,
                                         double* d_refs_out, int64_t* d_ref_obs_out, double* d_robust_mean_out,
                                         double* d_obs_desc_out) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && view && cfg && loss && d_refs_out && d_ref_obs_out, "pxr_ba_compute_references: NULL argument");
  PXR_REQUIRE(arena->C == 128 || arena->C == 64 || arena->C == 3 || arena->C == 1,
              "pxr_ba_compute_references: CHANNELS=%d not supported (128, 64, 3, 1)", arena->C);
  PXR_REQUIRE(iters >= 0, "pxr_ba_compute_references: negative iteration count");
  PXR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t n_obs = view->n_obs, n_pts = view->n_points;
  const int C = arena->C;
  if (n_pts == 0) return PXR_OK;
  // CSR of observations per point (host, structure only)
  std::vector<int32_t> obs_point(n_obs);
  PXR_HIP(hipMemcpyAsync(obs_point.data(), view->d_obs_point, 4 * n_obs, hipMemcpyDeviceToHost, st));
  PXR_HIP(hipStreamSynchronize(st));
  std::vector<int64_t> ptr(n_pts + 1, 0), lst(n_obs);
  for (int64_t i = 0; i < n_obs; ++i) {
    PXR_REQUIRE(obs_point[i] >= 0 && obs_point[i] < n_pts, "pxr_ba_compute_references: point index out of range");
    ++ptr[obs_point[i] + 1];
  }
  for (int64_t p = 0; p < n_pts; ++p) ptr[p + 1] += ptr[p];
  {
    std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < n_obs; ++i) lst[cur[obs_point[i]]++] = i;
  }
  // keep_observations (reference_extractor.h:60,259-265): the caller's buffer receives the per-observation
  // descriptors the IRLS runs on; otherwise they live in scratch
  double *d_desc = d_obs_desc_out, *d_desc_own = nullptr, *d_rec = nullptr, *d_w = nullptr;
  int64_t *d_ptr = nullptr, *d_lst = nullptr;
  auto cleanup = [&]() { (void)hipFree(d_desc_own); (void)hipFree(d_rec); (void)hipFree(d_w); (void)hipFree(d_ptr); (void)hipFree(d_lst); };
  const size_t no1 = n_obs ? (size_t)n_obs : 1;
  if ((!d_desc && hipMalloc((void**)&d_desc_own, sizeof(double) * no1 * C) != hipSuccess) ||
      hipMalloc((void**)&d_rec, sizeof(double) * no1 * PXR_OBS_REC) != hipSuccess ||
      hipMalloc((void**)&d_w, sizeof(double) * no1) != hipSuccess ||
      hipMalloc((void**)&d_ptr, sizeof(int64_t) * (n_pts + 1)) != hipSuccess ||
      hipMalloc((void**)&d_lst, sizeof(int64_t) * no1) != hipSuccess) {
    cleanup();
    return set_error(PXR_ENOMEM, "pxr_ba_compute_references: scratch allocation failed");
  }
  if (!d_desc) d_desc = d_desc_own;
  int rc = hip_check(hipMemcpyAsync(d_ptr, ptr.data(), sizeof(int64_t) * (n_pts + 1), hipMemcpyHostToDevice, st), "H2D");
  if (!rc && n_obs) rc = hip_check(hipMemcpyAsync(d_lst, lst.data(), sizeof(int64_t) * n_obs, hipMemcpyHostToDevice, st), "H2D");
  if (!rc && n_obs) {   // pass 1: descriptors at the current projections, no reference subtracted
    pxr_ba_view v = *view;
    v.d_refs = nullptr;
    rc = pxr_ba_eval(ctx, arena, &v, cfg, 0, d_rec, d_desc, nullptr, nullptr);
  }
  if (!rc) {
    const int G = C >= 64 ? 256 / (C / 8) : 256;
    const unsigned blocks = (unsigned)((n_pts + G - 1) / G);
    if (C == 3)
      hipLaunchKernelGGL((k_irls_small<3>), dim3(blocks), dim3(256), 0, st, n_pts, d_ptr, d_lst, d_desc, *loss, iters,
                         cfg->l2_normalize, d_w, d_refs_out, d_ref_obs_out, d_robust_mean_out);
    else if (C == 1)
      hipLaunchKernelGGL((k_irls_small<1>), dim3(blocks), dim3(256), 0, st, n_pts, d_ptr, d_lst, d_desc, *loss, iters,
                         cfg->l2_normalize, d_w, d_refs_out, d_ref_obs_out, d_robust_mean_out);
    else if (C == 128)
      hipLaunchKernelGGL((k_irls<128>), dim3(blocks), dim3(256), 0, st, n_pts, d_ptr, d_lst, d_desc, *loss, iters,
                         cfg->l2_normalize, d_w, d_refs_out, d_ref_obs_out, d_robust_mean_out);
    else
      hipLaunchKernelGGL((k_irls<64>), dim3(blocks), dim3(256), 0, st, n_pts, d_ptr, d_lst, d_desc, *loss, iters,
                         cfg->l2_normalize, d_w, d_refs_out, d_ref_obs_out, d_robust_mean_out);
    rc = hip_check(hipGetLastError(), "k_irls launch");
  }
  if (!rc) rc = hip_check(hipStreamSynchronize(st), "sync");
  cleanup();
  return rc;
}

// pxr_interp.h -- the per-observation interpolation core shared by the BA and KA kernels:
// 8 channels per lane, LPO lanes per patch evaluation (one DPP row for C = 128).
// BiCubicInterpolator::EvaluateSIMD (base/src/interpolation.h:177-218) + PixelInterpolator
// L2 normalisation (interpolation.h:642-677) in the reference's precision contract.
#pragma once
#include <hip/hip_runtime.h>

#include "pxr_device.h"

namespace pxr {

template <typename ST>
struct Texel8;  // 8 consecutive channels of one texel, widened for the horizontal pass

template <>
struct Texel8<_Float16> {
  typedef float work_t;
  uint4 raw;
  __device__ __forceinline__ void load(const _Float16* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void unpack(float out[8]) const {
    union { uint4 u; half8_t h; } cvt;
    cvt.u = raw;
    const float8_t f = __builtin_convertvector(cvt.h, float8_t);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = f[i];
  }
};
template <>
struct Texel8<float> {
  typedef float work_t;
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const float4*>(p);
    b = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ __forceinline__ void unpack(float out[8]) const {
    out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
    out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
  }
};
template <>
struct Texel8<double> {
  typedef double work_t;
  double2 a, b, c, d;
  __device__ __forceinline__ void load(const double* p) {
    a = *reinterpret_cast<const double2*>(p);
    b = *reinterpret_cast<const double2*>(p + 2);
    c = *reinterpret_cast<const double2*>(p + 4);
    d = *reinterpret_cast<const double2*>(p + 6);
  }
  __device__ __forceinline__ void unpack(double out[8]) const {
    out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y;
    out[4] = c.x; out[5] = c.y; out[6] = d.x; out[7] = d.y;
  }
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// floor(coordinate) as an int for the 4 x 4 stencil.  A coordinate far outside the patch (a diverging solver step: |x| > 2^31)
// makes the plain conversion undefined -- the compiled code then indexed memory with the garbage and the GPU raised a memory
// fault.  Every index below -3 selects the same clamped texels as -3, every index above n + 1 the same as n + 1 (Grid2D's
// clamping, grid2d.h:64-73), so the conversion is made on the clamped value; NaN lands on -3.
__device__ __forceinline__ int texel_index(double floored, int n) {
  return (int)fmin(fmax(floored, -3.0), (double)n + 1.0);
}

// ---- the three stages of BiCubicInterpolator::EvaluateSIMD (interpolation.h:177-218) for 8 channels of one patch ----
// Separate so that a kernel walking several channel chunks can issue the loads of the next chunk between the horizontal
// and the vertical pass of the current one (k_inner_packed); interp8_raw chains them.
struct StencilIndex {
  int ro[4], co[4];     // row offsets (in texels) and column indices of the 4 x 4 stencil, clamped (Grid2D, grid2d.h:64-73)
  double dy, dx;        // fractional position inside the centre texel
};
__device__ __forceinline__ StencilIndex stencil_index(int H, int W, double u, double v) {
  StencilIndex s;
  const double rf = floor(v), cf = floor(u);     // r = v (row), c = u (column)
  const int row = texel_index(rf, H), col = texel_index(cf, W);
  s.dy = v - rf; s.dx = u - cf;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s.ro[j] = clampi(row - 1 + j, 0, H - 1) * W;
    s.co[j] = clampi(col - 1 + j, 0, W - 1);
  }
  return s;
}
template <typename ST>
__device__ __forceinline__ void interp8_load(const ST* __restrict__ patch, int C, int chan0, const StencilIndex& si,
                                             Texel8<ST> (&tx)[4][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) tx[j][i].load(patch + (size_t)(si.ro[j] + si.co[i]) * C + chan0);
}
template <typename ST, bool WITH_JAC, bool FLOAT_SIMD>
__device__ __forceinline__ void interp8_horizontal(const Texel8<ST> (&tx)[4][4], double dx,
                                                   typename Texel8<ST>::work_t (&h)[4][8], typename Texel8<ST>::work_t (&hd)[4][8]) {
  typedef typename Texel8<ST>::work_t HT;  // horizontal-pass arithmetic type
  if constexpr (sizeof(HT) == 4) {
    const SplineCoefF32 kh(dx);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float p0[8], p1[8], p2[8], p3[8];
      tx[j][0].unpack(p0); tx[j][1].unpack(p1); tx[j][2].unpack(p2); tx[j][3].unpack(p3);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        float dd = 0.f;
        spline_f32<WITH_JAC>(p0[ch], p1[ch], p2[ch], p3[ch], kh, h[j][ch], dd);
        hd[j][ch] = dd;
      }
    }
  } else {
    const SplineCoefF64 kh(dx);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double p0[8], p1[8], p2[8], p3[8];
      tx[j][0].unpack(p0); tx[j][1].unpack(p1); tx[j][2].unpack(p2); tx[j][3].unpack(p3);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        double ff = 0, dd = 0;
        spline_f64<true, WITH_JAC>(p0[ch], p1[ch], p2[ch], p3[ch], kh, ff, dd);
        if (FLOAT_SIMD) { ff = (double)(float)ff; dd = (double)(float)dd; }
        h[j][ch] = ff; hd[j][ch] = dd;
      }
    }
  }
}
// frc (optional, WITH_JAC only): the cross derivative d2f / dr dc -- the derivative output of the vertical spline over
// the row derivatives; the reference leaves it un-normalised (interpolation.h:642-666)
template <typename HT, bool WITH_JAC, bool FLOAT_SIMD>
__device__ __forceinline__ void interp8_vertical(const HT (&h)[4][8], const HT (&hd)[4][8], double dy, double f[8], double fr[8],
                                                 double fc[8], double* frc = nullptr) {
  if constexpr (FLOAT_SIMD) {
    const SplineCoefF32 kv(dy);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      float ff, dd = 0.f, cc = 0.f, dummy = 0.f;
      spline_f32<WITH_JAC>((float)h[0][ch], (float)h[1][ch], (float)h[2][ch], (float)h[3][ch], kv, ff, dd);
      if (WITH_JAC) {
        if (frc) { spline_f32<true>((float)hd[0][ch], (float)hd[1][ch], (float)hd[2][ch], (float)hd[3][ch], kv, cc, dummy); frc[ch] = (double)dummy; }
        else spline_f32<false>((float)hd[0][ch], (float)hd[1][ch], (float)hd[2][ch], (float)hd[3][ch], kv, cc, dummy);
      }
      f[ch] = (double)ff; fr[ch] = (double)dd; fc[ch] = (double)cc;
    }
  } else {
    const SplineCoefF64 kv(dy);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      double ff = 0, dd = 0, cc = 0, dummy = 0;
      spline_f64<true, WITH_JAC>((double)h[0][ch], (double)h[1][ch], (double)h[2][ch], (double)h[3][ch], kv, ff, dd);
      if (WITH_JAC) {
        if (frc) { spline_f64<true, true>((double)hd[0][ch], (double)hd[1][ch], (double)hd[2][ch], (double)hd[3][ch], kv, cc, dummy); frc[ch] = dummy; }
        else spline_f64<true, false>((double)hd[0][ch], (double)hd[1][ch], (double)hd[2][ch], (double)hd[3][ch], kv, cc, dummy);
      }
      f[ch] = ff; fr[ch] = dd; fc[ch] = cc;
    }
  }
}

// Interpolated value + gradients of the 8 channels chan0 .. chan0 + 7 of one patch at (u, v), NOT normalised.
template <typename ST, bool WITH_JAC, bool FLOAT_SIMD>
__device__ __forceinline__ void interp8_raw(const ST* __restrict__ patch, int H, int W, int C, int chan0,
                                            double u, double v, double f[8], double fr[8], double fc[8],
                                            double* frc = nullptr) {
  const StencilIndex si = stencil_index(H, W, u, v);
  Texel8<ST> tx[4][4];
  interp8_load<ST>(patch, C, chan0, si, tx);
  typedef typename Texel8<ST>::work_t HT;
  HT h[4][8], hd[4][8];
  interp8_horizontal<ST, WITH_JAC, FLOAT_SIMD>(tx, si.dx, h, hd);
  interp8_vertical<HT, WITH_JAC, FLOAT_SIMD>(h, hd, si.dy, f, fr, fc, frc);
}

// Normalised descriptor + gradients of 8 channels of one observation, all lanes of the
// observation's lane group cooperating.  LPO = lanes per observation (C / 8).
template <typename ST, int LPO, bool WITH_JAC, bool FLOAT_SIMD>
__device__ __forceinline__ void interp8(const ST* __restrict__ patch, int H, int W, int C, int sub,
                                        double u, double v, bool l2_normalize, double f[8],
                                        double fr[8], double fc[8], double* frc = nullptr) {
  interp8_raw<ST, WITH_JAC, FLOAT_SIMD>(patch, H, W, C, sub * 8, u, v, f, fr, fc, frc);
  // PixelInterpolator::Evaluate L2 normalisation + chain rule, interpolation.h:648-666
  if (l2_normalize) {
    double ss = 0.0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ss = fma(f[ch], f[ch], ss);
    ss = (LPO == 16) ? row16_sum(ss) : row8_sum(ss);
    const double ninv = 1.0 / sqrt(ss);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) f[ch] *= ninv;
    if (WITH_JAC) {
      double dc = 0.0, dr = 0.0;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        fc[ch] *= ninv; fr[ch] *= ninv;
        dc = fma(f[ch], fc[ch], dc);
        dr = fma(f[ch], fr[ch], dr);
      }
      if (LPO == 16) { dc = row16_sum(dc); dr = row16_sum(dr); }
      else { dc = row8_sum(dc); dr = row8_sum(dr); }
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        fc[ch] = fma(-dc, f[ch], fc[ch]);
        fr[ch] = fma(-dr, f[ch], fr[ch]);
      }
    }
  }
}

// ---- value only, TWO 8-channel chunks per lane (channels 8 sub .. and 64 + 8 sub ..): a 128-channel descriptor on 8 lanes ----
// For passes that are chains of memory round trips rather than arithmetic (the line-search probes of the keypoint adjustment):
// 32 nodes per 256-thread trip instead of 16.  Same per-channel arithmetic as interp8; the L2 norm is summed over the 8 lanes.
template <typename ST, bool FLOAT_SIMD>
__device__ __forceinline__ void interp8x2_value(const ST* __restrict__ patch, int H, int W, int sub, double u, double v,
                                                bool l2_normalize, double fa[8], double fb[8]) {
  constexpr int C = 128;
  const double rf = floor(v), cf = floor(u);
  const int row = texel_index(rf, H), col = texel_index(cf, W);
  const double dy = v - rf, dx = u - cf;
  int ro[4], co[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ro[j] = clampi(row - 1 + j, 0, H - 1) * W;
    co[j] = clampi(col - 1 + j, 0, W - 1);
  }
  typedef typename Texel8<ST>::work_t HT;
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    double* f = part ? fb : fa;
    Texel8<ST> tx[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) tx[j][i].load(patch + (size_t)(ro[j] + co[i]) * C + part * 64 + sub * 8);
    HT h[4][8];
    if constexpr (sizeof(HT) == 4) {
      const SplineCoefF32 kh(dx);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float p0[8], p1[8], p2[8], p3[8];
        tx[j][0].unpack(p0); tx[j][1].unpack(p1); tx[j][2].unpack(p2); tx[j][3].unpack(p3);
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) { float dd = 0.f; spline_f32<false>(p0[ch], p1[ch], p2[ch], p3[ch], kh, h[j][ch], dd); }
      }
    } else {
      const SplineCoefF64 kh(dx);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double p0[8], p1[8], p2[8], p3[8];
        tx[j][0].unpack(p0); tx[j][1].unpack(p1); tx[j][2].unpack(p2); tx[j][3].unpack(p3);
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
          double ff = 0, dd = 0;
          spline_f64<true, false>(p0[ch], p1[ch], p2[ch], p3[ch], kh, ff, dd);
          if (FLOAT_SIMD) ff = (double)(float)ff;
          h[j][ch] = ff;
        }
      }
    }
    if constexpr (FLOAT_SIMD) {
      const SplineCoefF32 kv(dy);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        float ff, dd = 0.f;
        spline_f32<false>((float)h[0][ch], (float)h[1][ch], (float)h[2][ch], (float)h[3][ch], kv, ff, dd);
        f[ch] = (double)ff;
      }
    } else {
      const SplineCoefF64 kv(dy);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        double ff = 0, dd = 0;
        spline_f64<true, false>((double)h[0][ch], (double)h[1][ch], (double)h[2][ch], (double)h[3][ch], kv, ff, dd);
        f[ch] = ff;
      }
    }
  }
  if (l2_normalize) {
    double ss = 0.0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) ss = fma(fa[ch], fa[ch], fma(fb[ch], fb[ch], ss));
    ss = row8_sum(ss);
    const double ninv = 1.0 / sqrt(ss);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { fa[ch] *= ninv; fb[ch] *= ninv; }
  }
}

// ---- few-channel patches (cost maps: C = 1 or 3) ---------------------------------------------
// With fewer than 8 channels the reference leaves its SIMD path (base/src/interpolation.h:222-227) and runs
// [upstream] ceres::BiCubicInterpolator on doubles (:240-266): per row the Catmull-Rom spline
//   a = .5(-p0 + 3p1 - 3p2 + p3), b = .5(2p0 - 5p1 + 4p2 - p3), c = .5(-p0 + p2), d = p1,
//   f = d + x(c + x(b + x a)), f' = c + x(2b + 3a x)
// horizontally, then vertically on the four row values (f, df/dr) and on the four row derivatives (df/dc).
// One lane evaluates the whole texel; PixelInterpolator's L2 normalisation (:648-666) follows when asked for.
__device__ __forceinline__ void spline_ceres(double p0, double p1, double p2, double p3, double x, double& f,
                                             double& d) {
  const double a = 0.5 * (-p0 + 3.0 * p1 - 3.0 * p2 + p3);
  const double b = 0.5 * (2.0 * p0 - 5.0 * p1 + 4.0 * p2 - p3);
  const double c = 0.5 * (-p0 + p2);
  f = p1 + x * (c + x * (b + x * a));
  d = c + x * (2.0 * b + 3.0 * a * x);
}

template <typename ST, int C>
__device__ __forceinline__ void interp_small(const ST* __restrict__ patch, int H, int W, double u, double v,
                                             bool l2_normalize, double f[C], double fr[C], double fc[C]) {
  const double rf = floor(v), cf = floor(u);
  const int row = texel_index(rf, H), col = texel_index(cf, W);
  const double dy = v - rf, dx = u - cf;
  int co[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) co[i] = clampi(col - 1 + i, 0, W - 1) * C;   // Grid2D clamping, grid2d.h:64-73
  double h[4][C], hd[4][C];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const ST* rp = patch + (size_t)clampi(row - 1 + j, 0, H - 1) * W * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch)
      spline_ceres((double)rp[co[0] + ch], (double)rp[co[1] + ch], (double)rp[co[2] + ch], (double)rp[co[3] + ch], dx,
                   h[j][ch], hd[j][ch]);
  }
  double ss = 0.0;
#pragma unroll
  for (int ch = 0; ch < C; ++ch) {
    double dummy;
    spline_ceres(h[0][ch], h[1][ch], h[2][ch], h[3][ch], dy, f[ch], fr[ch]);
    spline_ceres(hd[0][ch], hd[1][ch], hd[2][ch], hd[3][ch], dy, fc[ch], dummy);
    ss += f[ch] * f[ch];
  }
  if (l2_normalize) {
    const double ninv = 1.0 / sqrt(ss);
    double dc = 0.0, dr = 0.0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      f[ch] *= ninv; fc[ch] *= ninv; fr[ch] *= ninv;
      dc += f[ch] * fc[ch]; dr += f[ch] * fr[ch];
    }
#pragma unroll
    for (int ch = 0; ch < C; ++ch) { fc[ch] -= dc * f[ch]; fr[ch] -= dr * f[ch]; }
  }
}

__device__ __forceinline__ double shfl_f64(double v, int src) {
  union { double d; int i[2]; } a;
  a.d = v;
  a.i[0] = __shfl(a.i[0], src);
  a.i[1] = __shfl(a.i[1], src);
  return a.d;
}
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src) {
  union { int64_t l; int i[2]; } a;
  a.l = v;
  a.i[0] = __shfl(a.i[0], src);
  a.i[1] = __shfl(a.i[1], src);
  return a.l;
}


}  // namespace pxr

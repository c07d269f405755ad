// pxr_gram.h -- the Gram-matrix form of the bicubic feature residual (shared by the inner iterations, pxr_ba_inner.hip, and the
// Gram-matrix cache of the LM loop, pxr_ba_gram.hip).
//
// Bicubic interpolation is linear in the 16 texels of the 4 x 4 stencil:  f = sum_t w_t T_t  with  w = wv (x) wu  (Catmull-Rom
// weights of the fractional position),  fc: wv (x) wu',  fr: wv' (x) wu.  With G = T T^t (16 x 16, contraction over the C
// channels) and D = T d (d: the reference descriptor) every channel sum the residual block's record needs is a quadratic or
// linear form in the weights:  f.f = w^t G w,  f.fc = w^t G wc, ...,  f.d = w.D  (base/src/interpolation.h:130-218 are the
// per-channel statements of the same sums).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "pxr_device.h"
#include "pxr_interp.h"

namespace pxr {

typedef double gd4 __attribute__((ext_vector_type(4)));

// Catmull-Rom weights of the four taps at fractional position x (the polynomial of CubicHermiteSpline, base/src/interpolation.h /
// cubic_hermite_spline_simd.h, expanded in the taps) and their derivatives
__device__ __forceinline__ void catmull_rom_weights(double x, double (&w)[4], double (&dw)[4]) {
  const double x2 = x * x, x3 = x2 * x;
  w[0] = -0.5 * x + x2 - 0.5 * x3; w[1] = 1.0 - 2.5 * x2 + 1.5 * x3; w[2] = 0.5 * x + 2.0 * x2 - 1.5 * x3; w[3] = -0.5 * x2 + 0.5 * x3;
  dw[0] = -0.5 + 2.0 * x - 1.5 * x2; dw[1] = -5.0 * x + 4.5 * x2; dw[2] = 0.5 + 4.0 * x - 4.5 * x2; dw[3] = -x + 1.5 * x2;
}
__device__ __forceinline__ double pick4(const double (&w)[4], int i) { return i == 0 ? w[0] : (i == 1 ? w[1] : (i == 2 ? w[2] : w[3])); }

// The texels of one 4 x 4 stencil as the MFMA wants them: lane (i = lane & 15, g = lane >> 4) holds C / 4 channels of texel i.
// WHICH channels is free -- the contraction runs over all of them, and step e of the MFMA chain only needs the four lanes
// g = 0..3 of every texel to hold the same four channels -- so request q of a lane takes the g-th 16-byte piece of the q-th
// 64-byte piece of its texel: the four lanes of a texel read 64 contiguous bytes per request (with a lane on C / 4 contiguous
// channels a request touched 16 bytes in every 64: twice the cache lines per instruction).  channel(g, e): the channel value(e)
// holds.  (row, col): the cell, wave-uniform.
template <typename ST, int C>
struct GramTexels {
  static constexpr int CPL = C / 4;
  static constexpr int VPL = sizeof(ST) == 2 ? 8 : 4;                 // values per 16-byte request
  typename std::conditional<sizeof(ST) == 2, uint4, float4>::type raw[CPL / VPL];
  static __device__ __forceinline__ int channel(int g, int e) { return (e / VPL) * (4 * VPL) + g * VPL + e % VPL; }
  __device__ __forceinline__ void load(const ST* __restrict__ patch, int H, int W, int row, int col) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    const int tr = clampi(row - 1 + (i >> 2), 0, H - 1), tc = clampi(col - 1 + (i & 3), 0, W - 1);   // Grid2D's clamp (grid2d.h:64-73)
    const ST* p = patch + ((size_t)tr * W + tc) * C + g * VPL;
    if constexpr (sizeof(ST) == 2) {
#pragma unroll
      for (int q = 0; q < CPL / 8; ++q) raw[q] = *reinterpret_cast<const uint4*>(p + 32 * q);
    } else {
#pragma unroll
      for (int q = 0; q < CPL / 4; ++q) raw[q] = *reinterpret_cast<const float4*>(p + 16 * q);
    }
  }
  __device__ __forceinline__ double value(int e) const {
    if constexpr (sizeof(ST) == 2) {
      union { uint4 u; _Float16 h[8]; } cv;
      cv.u = raw[e / 8];
      return (double)(float)cv.h[e % 8];
    } else {
      const float4 v = raw[e / 4];
      return (double)(e % 4 == 0 ? v.x : (e % 4 == 1 ? v.y : (e % 4 == 2 ? v.z : v.w)));
    }
  }
};

// The symmetric 16 x 16 Gram matrix in LDS as its ten upper 4 x 4 blocks (160 doubles instead of 256): block (R, Cb), R <= Cb, at
// gram_block(R, Cb) * 16, row-major inside.  Element (i, c) with i / 4 > c / 4 is read from the transposed block.
__device__ __forceinline__ constexpr int gram_block(int R, int Cb) { return R * 4 - R * (R - 1) / 2 + (Cb - R); }
constexpr int IG_GDOUBLES = 160;
// G = T T^t -> the blocked upper triangle at Gq, D = T d -> Dq[i], by ALL 64 lanes: the texel values are fed one per MFMA step as
// both operands (the channel order of the contraction is irrelevant); the accumulators come out as G[g + 4 r][i] (r = 0..3):
// in-block row g of block row r, column i.
template <typename ST, int C>
__device__ __forceinline__ void gram_contract(const GramTexels<ST, C>& tx, const double* ref, double* Gq, double* Dq) {
  constexpr int CPL = C / 4;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  gd4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  double dp = 0.0;
#pragma unroll
  for (int e = 0; e < CPL; e += 2) {
    const double x0 = tx.value(e), x1 = tx.value(e + 1);
#ifdef PXR_GRAM_PROBE_NO_MFMA      // tools/variant_build.sh: the build without its matrix instructions (what the rest costs)
    acc0[0] += x0; acc1[0] += x1;
#else
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc1, 0, 0, 0);
#endif
    dp = fma(x0, ref[GramTexels<ST, C>::channel(g, e)], dp);           // (e, e + 1: adjacent channels)
    dp = fma(x1, ref[GramTexels<ST, C>::channel(g, e + 1)], dp);
  }
  dp += __shfl_xor(dp, 16);
  dp += __shfl_xor(dp, 32);
  if (g == 0) Dq[i] = dp;
  const gd4 acc = acc0 + acc1;
  const int cb = i >> 2, ic = i & 3;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (cb >= r) Gq[gram_block(r, cb) * 16 + g * 4 + ic] = acc[r];
}

// rows i0 = 2 sub, i0 + 1 of G times the Kronecker-structured weights: (G w)_i, (G wc)_i, (G wr)_i for both rows
__device__ __forceinline__ void gram_rows_times_weights(const double* Gq, int sub, const double (&wu)[4], const double (&dwu)[4],
                                                        const double (&wv)[4], const double (&dwv)[4], double (&ya)[3], double (&yb)[3]) {
  const int R = sub >> 1, ir = 2 * (sub & 1);            // block row of both rows, in-block row of the first
  ya[0] = ya[1] = ya[2] = yb[0] = yb[1] = yb[2] = 0.0;
#pragma unroll
  for (int cbk = 0; cbk < 4; ++cbk) {                    // column block = the vertical tap r of the weights
    double ea[4], eb[4];
    if (cbk >= R) {                                       // stored block (R, cbk): two rows of four
      const double* blk = Gq + gram_block(R, cbk) * 16 + ir * 4;
      const double2 a01 = *reinterpret_cast<const double2*>(blk), a23 = *reinterpret_cast<const double2*>(blk + 2);
      const double2 b01 = *reinterpret_cast<const double2*>(blk + 4), b23 = *reinterpret_cast<const double2*>(blk + 6);
      ea[0] = a01.x; ea[1] = a01.y; ea[2] = a23.x; ea[3] = a23.y; eb[0] = b01.x; eb[1] = b01.y; eb[2] = b23.x; eb[3] = b23.y;
    } else {                                              // transposed block (cbk, R): two adjacent columns, rows 0..3
      const double* blk = Gq + gram_block(cbk, R) * 16 + ir;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const double2 t = *reinterpret_cast<const double2*>(blk + 4 * cc);
        ea[cc] = t.x; eb[cc] = t.y;
      }
    }
    const double za = fma(ea[3], wu[3], fma(ea[2], wu[2], fma(ea[1], wu[1], ea[0] * wu[0])));
    const double zb = fma(eb[3], wu[3], fma(eb[2], wu[2], fma(eb[1], wu[1], eb[0] * wu[0])));
    const double zca = fma(ea[3], dwu[3], fma(ea[2], dwu[2], fma(ea[1], dwu[1], ea[0] * dwu[0])));
    const double zcb = fma(eb[3], dwu[3], fma(eb[2], dwu[2], fma(eb[1], dwu[1], eb[0] * dwu[0])));
    ya[0] = fma(wv[cbk], za, ya[0]); yb[0] = fma(wv[cbk], zb, yb[0]);
    ya[1] = fma(wv[cbk], zca, ya[1]); yb[1] = fma(wv[cbk], zcb, yb[1]);
    ya[2] = fma(dwv[cbk], za, ya[2]); yb[2] = fma(dwv[cbk], zb, yb[2]);
  }
}

}  // namespace pxr

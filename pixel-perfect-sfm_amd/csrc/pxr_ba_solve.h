// pxr_ba_solve.h -- device-side structures shared by the BA solver's translation units
// (pxr_ba_solve.hip: linearisation, direct Schur + Cholesky path, LM loop; pxr_ba_pcg.hip: the
// iterative Schur path).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>

#include "pixsfm_hip.h"

namespace pxr {

struct SolveDev {          // device-side problem description shared by the kernels
  pxr_ba_view v;           // parameters being linearised (current or candidate)
  const int* pose_off; const int* pose_dim; const int* tmask;   // per image
  const int* intr_off; const int* intr_dim; const int* cmask;   // per camera
  const int* pt_var;                                             // per point
  const double* scale_c;   // [n_c]   Jacobi scaling, camera side
  const double* scale_p;   // [n_points][3]
  int n_c; int DC; int LS; // reduced system size, max camera-side columns, Lrec stride
  int ldS;                 // leading dimension of the reduced system buffer: n_c + 1 (rhs = last column)
};

struct ImgChunk { int img; int64_t begin, end; };   // observations [begin, end) of the image-ordered slot list

// global column index of camera-side column `a` of an observation in image img / camera cam
__device__ __forceinline__ int col_index(const SolveDev& d, int img, int cam, int a) {
  const int pd = d.pose_dim[img];
  return a < pd ? d.pose_off[img] + a : d.intr_off[cam] + (a - pd);
}

// Y_i = W_i T_p is formed on the fly where it is consumed (row a of observation i)
__device__ __forceinline__ void y_row(const double* __restrict__ W, const double* __restrict__ T, int64_t i, int a, int DC,
                                      int64_t pt, double& y0, double& y1, double& y2) {
  const double* Wi = W + ((size_t)i * DC + a) * 3;
  const double* Tp = T + 6 * (size_t)pt;
  const double w0 = Wi[0], w1 = Wi[1], w2 = Wi[2];
  y0 = w0 * Tp[0] + w1 * Tp[1] + w2 * Tp[2];
  y1 = w0 * Tp[1] + w1 * Tp[3] + w2 * Tp[4];
  y2 = w0 * Tp[2] + w1 * Tp[4] + w2 * Tp[5];
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

}  // namespace pxr

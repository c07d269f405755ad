// pxr_ba_solve.hip -- featuremetric bundle adjustment: Levenberg-Marquardt with Schur
// elimination of the point blocks, entirely on the GPU.
//
// Replaces BundleOptimizer::SolveProblem -> ceres::Solve
// (bundle_adjustment/src/bundle_optimizer.h:172-245) and the parameterisation of :335-453.
// [upstream Ceres 2.1] trust_region_minimizer.cc / levenberg_marquardt_strategy.cc semantics.
//
// Because every residual block's Jacobian factors as J = G P (G: C x 2 descriptor gradients,
// P: 2 x n projection Jacobian), the robustified normal equations of observation i are
//     J~^T J~ = P^T M~ P ,  J~^T r~ = P^T b~ ,   M~ = rho'(G^T G - kappa b b^T), b~ = rho' G^T r
// with the 2x2 / 2-vector (M~, b~) coming straight out of the fused record of pxr_ba_eval.
// The solver therefore never touches C-dimensional data: one pass of the HBM-bound residual
// kernel per LM iteration, everything else is per-observation 2 x (dc+3) algebra.
//
// Pipeline per linearisation (observations sorted by point for K1 locality):
//   K_jac     thread/obs    M~, b~, E = d(x,y)/dX (2x3), B = d(x,y)/d(pose,intr) tangent (2xdc) and
//             W_i = B^T M~ E (dc x 3); records assembled in LDS, written as coalesced streams
//   K_point   thread/point  V_p = sum E^T M~ E, g_p = sum E^T b~
//   K_img     block/(image,chunk)  U (pose/intrinsics blocks, upper) and g_c from LDS-staged records
// per LM attempt (radius changes on rejection, linearisation is reused):
//   K_pinv    thread/point  T_p = (V_p + D_p/radius)^-1
//   K_schur_lds  block/(image chunk, column tile)  S -= Y_i W_j^T over the point's observation pairs
//             (upper triangle only), rhs -= Y_i g_p, Y_i = W_i T_p formed on the fly; privatised in LDS
//   all-reduce(S | rhs) over ranks (RCCL, multi-GPU), + LM damping, blocked dense Cholesky (pxr_chol.hip)
//   K_backsub lane group/point  delta_p = -T_p (g_p + sum W_i^T delta_c)
//   K_update  x (+) delta (quaternion manifold, subset manifolds), then pxr_ba_eval at the
//             candidate WITH Jacobians (same HBM traffic as cost-only, saves the second
//             evaluation Ceres does after an accepted step).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "pxr_ba_pcg.h"
#include "pxr_ba_solve.h"
#include "pxr_device.h"
#include "pxr_internal.h"

namespace pxr {


// ---- K_jac ------------------------------------------------------------------------------------
// The 216-B L record and the 192-B W block of an observation are assembled in LDS (one row per lane)
// and written out by the wavefront as contiguous, fully coalesced streams: per-lane stores of
// 27 + 24 scattered doubles left partially written lines to be evicted and re-merged in HBM.
constexpr int JAC_THREADS = 128;

__global__ __launch_bounds__(JAC_THREADS) void k_jac(const SolveDev d, const double* __restrict__ rec,
                                                     pxr_loss loss, double* __restrict__ L, double* __restrict__ W) {
  extern __shared__ double jstage[];            // per wavefront: [64][LS] then [64][WS]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int LS = d.LS, W3 = 3 * d.DC, WS = W3 + 1;
  double* Lw = jstage + (size_t)wave * 64 * (LS + WS);
  double* Ww = Lw + (size_t)64 * LS;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + wave * 64;      // first observation of this wavefront
  const int64_t i = i0 + lane;
  const int n_valid = (int)min((int64_t)64, d.v.n_obs - i0);
  if (i < d.v.n_obs) {
  const int img = d.v.d_obs_image[i], pt = d.v.d_obs_point[i], cam = d.v.d_image_camera[img];
  double q[4], t[3], X[3], k[PXR_KPAD];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = d.v.d_qvec[4 * (size_t)img + j];
#pragma unroll
  for (int j = 0; j < 3; ++j) { t[j] = d.v.d_tvec[3 * (size_t)img + j]; X[j] = d.v.d_xyz[3 * (size_t)pt + j]; }
#pragma unroll
  for (int j = 0; j < PXR_KPAD; ++j) k[j] = d.v.d_cam_params[(size_t)cam * PXR_KPAD + j];
  const int model = d.v.d_cam_model[cam];
  double x, y, A[2][3], Pq[2][4], PX[2][3], Pk[2][PXR_KPAD];
  world_to_pixel_jac(model, k, q, t, X, x, y, A, Pq, PX, Pk);

  const double* r = rec + (size_t)i * PXR_OBS_REC;
  const double s = r[0], gxx = r[1], gxy = r[2], gyy = r[3], bx = r[4], by = r[5];
  double rho[3];
  loss_eval(loss.type, loss.a, 1.0, s, rho);
  // corrector [upstream Ceres corrector.cc]: J~^T J~ = rho'(J^T J - kappa (J^T r)(J^T r)^T)
  double kappa = 0.0;
  if (s != 0.0 && rho[2] > 0.0) {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - sqrt(D);
    kappa = (2.0 * alpha - alpha * alpha) / s;
  }
  double* Lo = Lw + (size_t)lane * LS;
  const double m00 = rho[1] * (gxx - kappa * bx * bx), m01 = rho[1] * (gxy - kappa * bx * by),
               m11 = rho[1] * (gyy - kappa * by * by);
  Lo[0] = m00; Lo[1] = m01; Lo[2] = m11;
  Lo[3] = rho[1] * bx;
  Lo[4] = rho[1] * by;
  // E = d(x,y)/dX, scaled
  const bool pvar = d.pt_var[pt] != 0;
  double e0[3], e1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const double sc = pvar ? d.scale_p[3 * (size_t)pt + j] : 0.0;
    e0[j] = PX[0][j] * sc; e1[j] = PX[1][j] * sc;
    Lo[5 + j] = e0[j];
    Lo[8 + j] = e1[j];
  }
  // W = B^T M~ E (dc x 3) is fixed for the whole linearisation point (it does not depend on the
  // trust-region radius), so it is produced here once instead of once per LM attempt
  double me0[3], me1[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    me0[j] = m00 * e0[j] + m01 * e1[j];
    me1[j] = m01 * e0[j] + m11 * e1[j];
  }
  double* Wo = Ww + (size_t)lane * WS;
  // B: tangent pose columns then variable intrinsics columns
  double* B0 = Lo + 11;
  double* B1 = B0 + d.DC;
  int col = 0;
  auto put = [&](int c, double b0, double b1) {
    B0[c] = b0; B1[c] = b1;
    Wo[3 * c] = b0 * me0[0] + b1 * me1[0];
    Wo[3 * c + 1] = b0 * me0[1] + b1 * me1[1];
    Wo[3 * c + 2] = b0 * me0[2] + b1 * me1[2];
  };
  if (d.pose_dim[img] > 0) {
    const int po = d.pose_off[img];
    // QuaternionManifold::PlusJacobian [upstream Ceres manifold.cc], 4x3
    const double PJ[4][3] = {{-q[1], -q[2], -q[3]}, {q[0], q[3], -q[2]}, {-q[3], q[0], q[1]}, {q[2], -q[1], q[0]}};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double sc = d.scale_c[po + a];
      put(col, (Pq[0][0] * PJ[0][a] + Pq[0][1] * PJ[1][a] + Pq[0][2] * PJ[2][a] + Pq[0][3] * PJ[3][a]) * sc,
          (Pq[1][0] * PJ[0][a] + Pq[1][1] * PJ[1][a] + Pq[1][2] * PJ[2][a] + Pq[1][3] * PJ[3][a]) * sc);
      ++col;
    }
    const int tm = d.tmask[img];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if ((tm >> a) & 1) continue;
      const double sc = d.scale_c[po + col];
      put(col, A[0][a] * sc, A[1][a] * sc);
      ++col;
    }
  }
  if (d.intr_dim[cam] > 0) {
    const int io = d.intr_off[cam], cm = d.cmask[cam], K = camera_num_params(model);
    int kc = 0;
#pragma unroll
    for (int a = 0; a < PXR_KPAD; ++a) {
      if (a >= K || ((cm >> a) & 1)) continue;
      const double sc = d.scale_c[io + kc];
      put(col, Pk[0][a] * sc, Pk[1][a] * sc);
      ++col; ++kc;
    }
  }
  for (; col < d.DC; ++col) put(col, 0.0, 0.0);
  }
  __threadfence_block();                        // LDS rows of the other lanes of this wavefront
  __builtin_amdgcn_wave_barrier();
  if (n_valid <= 0) return;
  double* Lg = L + (size_t)i0 * LS;
  for (int x = lane; x < n_valid * LS; x += 64) Lg[x] = Lw[x];
  double* Wg = W + (size_t)i0 * W3;
  for (int x = lane; x < n_valid * W3; x += 64) { const int o = x / W3; Wg[x] = Ww[o * WS + (x - o * W3)]; }
}

// ---- K_point ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_point(const SolveDev d, const int64_t* __restrict__ pt_ptr,
                                               const int64_t* __restrict__ pt_obs,
                                               const double* __restrict__ L, double* __restrict__ V,
                                               double* __restrict__ gp) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= d.v.n_points) return;
  double v[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (int64_t o = pt_ptr[p]; o < pt_ptr[p + 1]; ++o) {
    const double* Lo = L + (size_t)pt_obs[o] * d.LS;
    const double m00 = Lo[0], m01 = Lo[1], m11 = Lo[2], b0 = Lo[3], b1 = Lo[4];
    const double e0[3] = {Lo[5], Lo[6], Lo[7]}, e1[3] = {Lo[8], Lo[9], Lo[10]};
    double me0[3], me1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { me0[j] = m00 * e0[j] + m01 * e1[j]; me1[j] = m01 * e0[j] + m11 * e1[j]; }
    v[0] += e0[0] * me0[0] + e1[0] * me1[0];
    v[1] += e0[0] * me0[1] + e1[0] * me1[1];
    v[2] += e0[0] * me0[2] + e1[0] * me1[2];
    v[3] += e0[1] * me0[1] + e1[1] * me1[1];
    v[4] += e0[1] * me0[2] + e1[1] * me1[2];
    v[5] += e0[2] * me0[2] + e1[2] * me1[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j] += e0[j] * b0 + e1[j] * b1;
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) V[6 * p + j] = v[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) gp[3 * p + j] = g[j];
}

// ---- K_img: U and g_c --------------------------------------------------------------------------
constexpr int IMG_BATCH = 128;   // observations staged in LDS per pass

__global__ __launch_bounds__(256) void k_img(const SolveDev d, const ImgChunk* __restrict__ chunks,
                                             const int64_t* __restrict__ img_obs,
                                             const double* __restrict__ L, double* __restrict__ U,
                                             double* __restrict__ gc, int block_form, double det_scale,
                                             double* __restrict__ chunk_trace, double* __restrict__ kpart, int ne_max) {
  // kpart (deterministic mode of the iterative solver): the chunk's NE sums go to kpart[chunk][ne_max] instead of into U and g_c by
  // atomics; pcg_blocks_from_partials adds them per image and per column in a fixed order
  // block_form = 0: U is the dense n_c x n_c matrix (upper triangle), direct solver;
  // block_form = 1: U is [n_images][DC][DC], the image's full symmetric dc x dc block (iterative solver)
  extern __shared__ double stage[];            // [IMG_BATCH][LS] records, then [256] reduction slots
  const ImgChunk ch = chunks[blockIdx.x];
  const int img = ch.img, cam = d.v.d_image_camera[img];
  const int dc = d.pose_dim[img] + d.intr_dim[cam];
  if (dc == 0) { if (threadIdx.x == 0 && chunk_trace) chunk_trace[blockIdx.x] = 0.0; return; }
  const int LS = d.LS;
  double* red = stage + (size_t)IMG_BATCH * LS;
  const int NP = dc * (dc + 1) / 2, NE = NP + dc;
  const int slices = 256 / NE;                 // NE <= 189
  const int e = threadIdx.x % NE, sl = threadIdx.x / NE;
  int a = 0, b = 0;
  const bool is_g = e >= NP;
  if (is_g) { a = e - NP; }
  else { int rem = e; a = 0; while (rem >= dc - a) { rem -= dc - a; ++a; } b = a + rem; }
  double acc = 0.0;
  // deterministic mode: every OBSERVATION's term is rounded to the fixed-point grid on its own and the integers are added (`qacc`)
  // -- a chunk's double sum rounded once would depend on which observations share a chunk, i.e. on how the points are dealt
  // to ranks; `acc` (the plain double sum) then only feeds the trace check of the overflow guard (chunk_trace)
  long long qacc = 0;
  const bool fixed = det_scale != 0.0;
  // The records of an image's observations are gathered (216 B each at DC = 8) into LDS with
  // coalesced, independent loads -- consecutive lanes read consecutive doubles of a record -- and
  // every thread then accumulates its matrix entry from LDS.  (A per-thread gather loop over the
  // observations was bound by the latency of its dependent loads.)
  // (the next batch's records are requested into registers before the current batch is consumed: a load -> LDS -> accumulate
  // loop exposed the gather's latency once per batch)
  constexpr int IMG_PF = 16;                   // doubles per thread and batch: IMG_BATCH * LS <= 256 * IMG_PF (LS <= 32)
  double pf[IMG_PF];
  auto request = [&](int64_t base) {
    const int nb = (int)min((int64_t)IMG_BATCH, ch.end - base);
#pragma unroll
    for (int j = 0; j < IMG_PF; ++j) {
      const int x = threadIdx.x + 256 * j;
      const int xc = min(x, nb * LS - 1);
      const int o = xc / LS, f = xc - o * LS;
      pf[j] = L[(size_t)img_obs[base + o] * LS + f];
    }
  };
  if (LS <= 2 * IMG_PF) {
    if (ch.begin < ch.end) request(ch.begin);
    for (int64_t base = ch.begin; base < ch.end; base += IMG_BATCH) {
      const int nb = (int)min((int64_t)IMG_BATCH, ch.end - base);
#pragma unroll
      for (int j = 0; j < IMG_PF; ++j) { const int x = threadIdx.x + 256 * j; if (x < nb * LS) stage[x] = pf[j]; }
      __syncthreads();
      if (base + IMG_BATCH < ch.end) request(base + IMG_BATCH);
      if (sl < slices) {
        for (int o = sl; o < nb; o += slices) {
          const double* Lo = stage + (size_t)o * LS;
          const double* B0 = Lo + 11; const double* B1 = B0 + d.DC;
          const double term = is_g ? B0[a] * Lo[3] + B1[a] * Lo[4]
                                   : B0[a] * (Lo[0] * B0[b] + Lo[1] * B1[b]) + B1[a] * (Lo[1] * B0[b] + Lo[2] * B1[b]);
          acc += term;
          if (fixed) qacc += __double2ll_rn(term * det_scale);
        }
      }
      __syncthreads();
    }
  } else {
  for (int64_t base = ch.begin; base < ch.end; base += IMG_BATCH) {
    const int nb = (int)min((int64_t)IMG_BATCH, ch.end - base);
    for (int x = threadIdx.x; x < nb * LS; x += 256) {
      const int o = x / LS, f = x - o * LS;
      stage[x] = L[(size_t)img_obs[base + o] * LS + f];
    }
    __syncthreads();
    if (sl < slices) {
      for (int o = sl; o < nb; o += slices) {
        const double* Lo = stage + (size_t)o * LS;
        const double* B0 = Lo + 11; const double* B1 = B0 + d.DC;
        const double term = is_g ? B0[a] * Lo[3] + B1[a] * Lo[4]
                                 : B0[a] * (Lo[0] * B0[b] + Lo[1] * B1[b]) + B1[a] * (Lo[1] * B0[b] + Lo[2] * B1[b]);
        acc += term;
        if (fixed) qacc += __double2ll_rn(term * det_scale);
      }
    }
    __syncthreads();
  }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0) for (int s2 = 1; s2 < slices; ++s2) acc += red[s2 * NE + e];
  if (fixed) {                                   // the integers' turn through the same reduction slots
    __syncthreads();
    red[threadIdx.x] = __longlong_as_double(qacc);
    __syncthreads();
    if (sl == 0) for (int s2 = 1; s2 < slices; ++s2) qacc += __double_as_longlong(red[s2 * NE + e]);
    __syncthreads();
    // the chunk's trace(U) contribution, in plain doubles: only compared with a threshold (overflow guard of the solver)
    red[threadIdx.x] = (sl == 0 && !is_g && a == b) ? acc : 0.0;
    __syncthreads();
    if (threadIdx.x == 0 && chunk_trace) {
      double tr = 0.0;
      for (int x = 0; x < NP; ++x) tr += red[x];
      chunk_trace[blockIdx.x] = tr;
    }
  }
  if (sl == 0 && kpart) {
    kpart[(size_t)blockIdx.x * ne_max + e] = acc;
  } else if (sl == 0) {
    const int ra = col_index(d, img, cam, a);
    // (det_scale != 0: deterministic mode, order-independent fixed-point accumulation -- pxr_device.h)
    const double v = fixed ? __longlong_as_double(qacc) : acc;
    if (is_g) accum_flush(gc + ra, v, det_scale);
    else if (block_form) {
      double* Ub = U + (size_t)img * d.DC * d.DC;
      accum_flush(Ub + a * d.DC + b, v, det_scale);
      if (a != b) accum_flush(Ub + b * d.DC + a, v, det_scale);
    } else accum_flush(U + (size_t)ra * d.n_c + col_index(d, img, cam, b), v, det_scale);
  }
}

// ---- K_pinv: T_p = (V_p + D_p / radius)^-1 -----------------------------------------------------
__global__ __launch_bounds__(256) void k_pinv(int64_t n_points, const int* __restrict__ pt_var,
                                              const double* __restrict__ V,
                                              const double* __restrict__ Vdiag0 /* clamped diag */,
                                              double inv_radius, double* __restrict__ T,
                                              double* __restrict__ zero_ptr, int zero_n) {
  // side job of the attempt's first kernel: clear the attempt's scalars / limbs / info word (one fill launch less per attempt;
  // nothing before the Schur kernels of this attempt touches them)
  if (blockIdx.x == 0) for (int e = threadIdx.x; e < zero_n; e += blockDim.x) zero_ptr[e] = 0.0;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  double* To = T + 6 * p;
  if (!pt_var[p]) { for (int j = 0; j < 6; ++j) To[j] = 0.0; return; }
  const double a00 = V[6 * p] + Vdiag0[3 * p] * inv_radius, a01 = V[6 * p + 1], a02 = V[6 * p + 2];
  const double a11 = V[6 * p + 3] + Vdiag0[3 * p + 1] * inv_radius, a12 = V[6 * p + 4];
  const double a22 = V[6 * p + 5] + Vdiag0[3 * p + 2] * inv_radius;
  const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double id = 1.0 / det;
  To[0] = c00 * id; To[1] = c01 * id; To[2] = c02 * id;
  To[3] = (a00 * a22 - a02 * a02) * id; To[4] = (a01 * a02 - a00 * a12) * id;
  To[5] = (a00 * a11 - a01 * a01) * id;
}

// the accepted candidate becomes the current point: four arrays, ONE launch (four hipMemcpyAsync cost four dispatches)
struct Copy4 { const double* src[4]; double* dst[4]; int64_t n[4]; };
__global__ __launch_bounds__(256) void k_copy4(const Copy4 c) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int a = 0; a < 4; ++a)
    for (int64_t e = t0; e < c.n[a]; e += stride) c.dst[a][e] = c.src[a][e];
}

// ---- K_schur: S -= Y_i W_j^T (upper), rhs -= Y_i g_p ----------------------------------------------
__global__ __launch_bounds__(256) void k_schur(const SolveDev d, const int64_t* __restrict__ pt_ptr,
                                               const int64_t* __restrict__ pt_obs,
                                               const double* __restrict__ W, const double* __restrict__ T,
                                               const double* __restrict__ gp, double* __restrict__ S,
                                               double* __restrict__ rhs, double det_scale) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = tid / d.DC;
  const int a = (int)(tid - i * d.DC);
  if (i >= d.v.n_obs) return;
  const int img = d.v.d_obs_image[i], cam = d.v.d_image_camera[img];
  const int dci = d.pose_dim[img] + d.intr_dim[cam];
  if (a >= dci) return;
  const int pt = d.v.d_obs_point[i];
  if (!d.pt_var[pt]) return;
  double y0, y1, y2;
  y_row(W, T, i, a, d.DC, pt, y0, y1, y2);
  const int r = col_index(d, img, cam, a);
  accum_add(rhs + (size_t)r * d.ldS, -(y0 * gp[3 * (size_t)pt] + y1 * gp[3 * (size_t)pt + 1] + y2 * gp[3 * (size_t)pt + 2]), det_scale);
  for (int64_t o = pt_ptr[pt]; o < pt_ptr[pt + 1]; ++o) {
    const int64_t j = pt_obs[o];
    const int imgj = d.v.d_obs_image[j], camj = d.v.d_image_camera[imgj];
    const int dcj = d.pose_dim[imgj] + d.intr_dim[camj];
    const double* Wj = W + (size_t)j * d.DC * 3;
    for (int b = 0; b < dcj; ++b) {
      const int c = col_index(d, imgj, camj, b);
      if (r > c) continue;   // upper triangle only: the mirrored entry comes from the ordered pair (j, i)
      accum_add(S + (size_t)r * d.ldS + c, -(y0 * Wj[3 * b] + y1 * Wj[3 * b + 1] + y2 * Wj[3 * b + 2]), det_scale);
    }
  }
}

// ---- K_schur_lds: same contraction, privatised in LDS ------------------------------------------------
// A workgroup owns (image chunk, column tile): it accumulates the block row
//   S[rows of (pose(img), intr(cam)), c0 : c0 + CT]  -=  Y_i W_j^T
// in LDS with ds_add_f64 and flushes it once with global atomics.  The dense-S atomics drop
// from one per (i, j, a, b) contribution (~1.7e8 at 1M observations) to one per (chunk, row,
// column) (~2.5e6); the contraction itself runs at LDS-atomic speed.
template <int G>   // lanes per observation: G >= DC (8, 16 or 32)
__global__ __launch_bounds__(1024) void k_schur_lds(const SolveDev d, const ImgChunk* __restrict__ chunks,
                                                    const int4* __restrict__ so,      // per image-ordered slot: {obs, point, first partner, partners (0 = constant point)}
                                                    const int* __restrict__ pj,       // partner observation of each (point, k) slot
                                                    const int4* __restrict__ pcols,   // its {pose_off, pose_dim, intr_off, intr_dim}
                                                    const double* __restrict__ W, const double* __restrict__ T,
                                                    const double* __restrict__ gp, int CT,
                                                    double* __restrict__ S, double* __restrict__ rhs, double det_scale) {
  extern __shared__ double acc[];              // [DC][CT] then [DC] for the right-hand side
  const ImgChunk ch = chunks[blockIdx.x];
  const int img = ch.img, cam = d.v.d_image_camera[img];
  const int dci = d.pose_dim[img] + d.intr_dim[cam];
  if (dci == 0) return;
  const int c0 = blockIdx.y * CT, c1 = min(d.n_c, c0 + CT);
  double* racc = acc + (size_t)d.DC * CT;
  double* ybuf = racc + d.DC;                  // [blockDim.x][3]: the rows of Y_i, handed round the lane group through LDS
  for (int e = threadIdx.x; e < dci * CT + d.DC; e += blockDim.x) {
    if (e < dci * CT) acc[(e / CT) * CT + (e % CT)] = 0.0; else racc[e - dci * CT] = 0.0;
  }
  __syncthreads();
  // G lanes share one observation.  Lane a first forms row a of Y_i = W_i T_p; the dc_i x 3 rows are
  // handed round the group ONCE per observation (shuffles).  Then, per partner observation j of the
  // same point, lane b loads row b of W_j (one coalesced read of the whole dc_j x 3 block per group,
  // all partners of a batch in flight together) and adds its column of the block -Y_i W_j^T to the
  // LDS tile.  The host flattens obs -> point -> partner list -> column descriptors into three
  // levels of loads (so[], then pj / pcols / W_i / T / gp, then W_j).
  const int lane_b = threadIdx.x % G, grp = threadIdx.x / G, n_grp = blockDim.x / G;
  int rrow[G];                                             // global row of Y_i row a (same for every observation of the image)
#pragma unroll
  for (int a = 0; a < G; ++a) rrow[a] = a < dci ? col_index(d, img, cam, a) : 0x7fffffff;
  // The loads of an observation form a chain: slot descriptor -> (W_i, T_p, g_p, the partner list) -> the partners' rows.  As
  // first written every link waited for the one before it -- and W_i / T, g_p, the partner descriptors were three links of
  // their own: five dependent round trips per observation, eight observations per lane group, 80 % of the wave cycles
  // waiting (profiles/r4_hot_kernels_pmc.json).  Now: the next observation's descriptor is requested one iteration ahead,
  // everything it addresses goes out TOGETHER (unconditional loads at clamped indices, selected afterwards: a conditional
  // load is a branch + a wait), then the partners' rows: two links on the critical path.
  constexpr int PB = 8;                                    // partners per batch of loads in flight
  const bool first_tile = blockIdx.y == 0;
  int64_t o = ch.begin + grp;
  int4 s = o < ch.end ? so[o] : make_int4(0, 0, 0, 0);
  while (o < ch.end) {
    const int64_t o_next = o + n_grp;
    const int4 s_next = o_next < ch.end ? so[o_next] : make_int4(0, 0, 0, 0);
    if (s.w != 0) {                                        // (0: constant point; uniform over the group)
      const int64_t i = s.x, pt = s.y;
      const bool row_ok = lane_b < dci;
      // -- link 1: W_i row, T_p, g_p, partner observations and their column descriptors
      const double* Wi = W + ((size_t)i * d.DC + (row_ok ? lane_b : 0)) * 3;
      const double* Tp = T + 6 * (size_t)pt;
      const double* gpp = gp + 3 * (size_t)pt;
      const double w0 = Wi[0], w1 = Wi[1], w2 = Wi[2];
      const double t0 = Tp[0], t1 = Tp[1], t2 = Tp[2], t3 = Tp[3], t4 = Tp[4], t5 = Tp[5];
      const double g0 = gpp[0], g1 = gpp[1], g2 = gpp[2];
      double y0 = 0.0, y1 = 0.0, y2 = 0.0;
      // (one copy of the batch body: tracks of more than PB observations -- rare -- come round again with kb = PB, 2 PB, ...)
      for (int kb = 0; kb < s.w; kb += PB) {
        int jj[PB];
        int4 cc[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int idx = s.z + min(kb + u, s.w - 1);
          jj[u] = pj[idx]; cc[u] = pcols[idx];
        }
        // -- link 2: row lane_b of every partner's W_j
        double mm[PB][3];
        int col[PB];                                        // global column of that row, or -1: not this lane's / outside the tile
#pragma unroll
        for (int u = 0; u < PB; ++u) {
          const int pdj = cc[u].y, dcj = pdj + cc[u].w;
          const bool has = kb + u < s.w && lane_b < dcj;
          const double* Wj = W + ((size_t)jj[u] * d.DC + (has ? lane_b : 0)) * 3;
          mm[u][0] = Wj[0]; mm[u][1] = Wj[1]; mm[u][2] = Wj[2];
          const int c = lane_b < pdj ? cc[u].x + lane_b : cc[u].z + (lane_b - pdj);
          col[u] = (has && c >= c0 && c < c1) ? c : -1;
        }
        if (kb == 0) {     // -- Y_i row (from link 1) and the right-hand side, while the partners' rows are on their way
          if (row_ok) { y0 = w0 * t0 + w1 * t1 + w2 * t2; y1 = w0 * t1 + w1 * t3 + w2 * t4; y2 = w0 * t2 + w1 * t4 + w2 * t5; }
          if (first_tile && row_ok) accum_add(racc + lane_b, -(y0 * g0 + y1 * g1 + y2 * g2), det_scale);
          // (a wavefront's LDS operations execute in order: the group's reads below see these writes, and the next
          // observation's writes come after this one's reads)
          double* yb = ybuf + 3 * (size_t)threadIdx.x;
          yb[0] = y0; yb[1] = y1; yb[2] = y2;
        }
        // row a of Y_i comes round the group through LDS (three broadcast reads: all lanes of the group read one address; as
        // 64-bit shuffles they were six ds_bpermute per row) and meets every partner's row: the 24 values of all rows at once
        // were 48 registers next to the 48 of the partners' rows
        const double* yg = ybuf + 3 * (size_t)(threadIdx.x - lane_b);
#pragma unroll
        for (int a = 0; a < G; ++a) {
          const double ya0 = yg[3 * a], ya1 = yg[3 * a + 1], ya2 = yg[3 * a + 2];
#pragma unroll
          for (int u = 0; u < PB; ++u) {
            const int c = col[u];
            if (c >= 0 && c >= rrow[a])                      // upper triangle (rrow = INT_MAX beyond dc_i)
              accum_add(acc + (size_t)a * CT + (c - c0), -(ya0 * mm[u][0] + ya1 * mm[u][1] + ya2 * mm[u][2]), det_scale);
          }
        }
      }
    }
    o = o_next; s = s_next;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < dci * CT; e += blockDim.x) {
    const int a = e / CT, cl = e % CT;
    const double v = acc[(size_t)a * CT + cl];
    if (__double_as_longlong(v) != 0ll) accum_flush(S + (size_t)col_index(d, img, cam, a) * d.ldS + (c0 + cl), v, det_scale);
  }
  if (blockIdx.y == 0 && threadIdx.x < dci) {
    const double v = racc[threadIdx.x];
    if (__double_as_longlong(v) != 0ll) accum_flush(rhs + (size_t)col_index(d, img, cam, threadIdx.x) * d.ldS, v, det_scale);
  }
}

// ---- small vector kernels --------------------------------------------------------------------------
// S is n x (n + 1) row-major (column n = right-hand side).  add_u = 1: S <- [upper(U) | 0];
// add_u = 0: S += diag(damp) / radius and rhs += gc (after the all-reduce).
__global__ void k_copy_upper_add_diag(int n, const double* __restrict__ U, const double* __restrict__ damp,
                                      double inv_radius, const double* __restrict__ gc, double* __restrict__ S,
                                      int add_u, double det_scale) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int ld = n + 1;
  if (t >= (int64_t)n * ld) return;
  const int r = (int)(t / ld), c = (int)(t % ld);
  double v = S[t];
  if (add_u) {
    // deterministic mode: U still holds its fixed-point integers (the same grid), the Schur kernel adds integers to these slots and
    // the second pass of this kernel (add_u = 0) turns the sums back -- AFTER the ranks' integers were added: a slot of 2^59 units does not survive a round
    // trip through a double (53 bits), and rounding every rank's partial sum would make the result depend on the partition
    v = (c < n && r <= c) ? U[(size_t)r * n + c] : 0.0;
  }
  else {
    if (det_scale != 0.0) v = (double)__double_as_longlong(v) / det_scale;    // (the slots' integers become doubles here: one pass)
    if (r == c) v += damp[r] * inv_radius;
    if (c == n) v += gc[r];
  }
  S[t] = v;
}

__global__ void k_extract_diag(int n, const double* __restrict__ U, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = U[(size_t)i * n + i];
}

__global__ void k_clamp(int64_t n, const double* __restrict__ in, double lo, double hi, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = fmin(fmax(in[i], lo), hi);
}

__global__ void k_point_diag(int64_t n_points, const double* __restrict__ V, double* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_points) return;
  out[3 * p] = V[6 * p]; out[3 * p + 1] = V[6 * p + 3]; out[3 * p + 2] = V[6 * p + 5];
}

__global__ void k_jacobi_scale(int64_t n, const double* __restrict__ diag, double* __restrict__ scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scale[i] = 1.0 / (1.0 + sqrt(diag[i]));
}

__global__ void k_axpy1(int n, const double* __restrict__ x, double* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

__global__ void k_fill(int64_t n, double v, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

// delta_c = -x ; accumulates replicated scalars: [0] d.D.d - d.g (camera side)  [4] max|g_c/scale|
__global__ void k_finish_camera_step(int n, const double* __restrict__ x, const double* __restrict__ gc,
                                     const double* __restrict__ damp, double inv_radius,
                                     double* __restrict__ delta_c, double* __restrict__ scal_rep, long long* __restrict__ limb) {
  __shared__ long long lsh[PXR_LIMBS * 4];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double part = 0.0;
  if (i < n) {
    const double dl = -x[i];
    delta_c[i] = dl;
    part = dl * (damp[i] * inv_radius * dl - gc[i]);
  }
  if (limb) {                                    // deterministic mode: one limb addend per column (pxr_device.h)
    Limbs l;
    if (i < n) l.add(part);
    limbs_block_add(l, limb + (8 + 0) * PXR_LIMBS, lsh);
    return;
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if ((threadIdx.x & 63) == 0) atomicAdd(scal_rep + 0, part);
}

// ---- K_backsub: delta_p = -T (g_p + sum_i W_i^T delta_c) ---------------------------------------------
// G lanes share one point (lane a = row a of every W_i: one coalesced 24 G-byte read per observation and group; a thread per
// point read its 24 doubles per observation at the stride of the whole track and waited on each -- 136 us at 1M observations
// against 45 us for the same layout in k_pt_u of the iterative solver).  Fixed butterfly over the group, lane 0 applies T_p.
constexpr int BACKSUB_PPG = 8;
template <int G>
__global__ __launch_bounds__(256) void k_backsub(const SolveDev d, const int64_t* __restrict__ pt_ptr,
                                                 const int* __restrict__ pj, const int4* __restrict__ pcols,
                                                 const double* __restrict__ W, const double* __restrict__ T,
                                                 const double* __restrict__ gp, const double* __restrict__ delta_c,
                                                 const double* __restrict__ Vdiag0, double inv_radius,
                                                 double* __restrict__ delta_p, double* __restrict__ scal_sum, long long* __restrict__ limb) {
  __shared__ double red[256 / 64];
  __shared__ long long lsh[PXR_LIMBS * 4];
  const int lane_a = threadIdx.x % G;
  double part = 0.0;
  Limbs lpart;                                   // deterministic mode: one limb addend per POINT
  // BACKSUB_PPG points per group and one atomic per workgroup: the model-cost scalar is ONE address, and an atomic per
  // wavefront of eight points (25 000 of them) serialised there for longer than the kernel's own work
  for (int rep = 0; rep < BACKSUB_PPG; ++rep) {
  const int64_t p = ((int64_t)blockIdx.x * BACKSUB_PPG + rep) * (256 / G) + threadIdx.x / G;
  const bool live = p < d.v.n_points;                  // whole groups are live or not
  const bool var = live && d.pt_var[p];
  double v0 = 0.0, v1 = 0.0, v2 = 0.0;
  if (var) {
    // four observations per trip: their descriptors first, then every W row and camera-step entry they address, at clamped
    // indices (one observation per trip was a chain of two dependent gathers per observation); the additions keep the track's order
    const int64_t o_end = pt_ptr[p + 1];
    for (int64_t o = pt_ptr[p]; o < o_end; o += 4) {
      int4 ci[4];                                        // {pose_off, pose_dim, intr_off, intr_dim} of the observation pj[.]
      int jj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int64_t idx = min(o + u, o_end - 1); ci[u] = pcols[idx]; jj[u] = pj[idx]; }
      double w[4][3], dcv[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ok[u] = o + u < o_end && lane_a < ci[u].y + ci[u].w;
        const double* Wi = W + ((size_t)jj[u] * d.DC + (ok[u] ? lane_a : 0)) * 3;
        w[u][0] = Wi[0]; w[u][1] = Wi[1]; w[u][2] = Wi[2];
        dcv[u] = delta_c[ok[u] ? (lane_a < ci[u].y ? ci[u].x + lane_a : ci[u].z + (lane_a - ci[u].y)) : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (ok[u]) { v0 += w[u][0] * dcv[u]; v1 += w[u][1] * dcv[u]; v2 += w[u][2] * dcv[u]; }
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {
    v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off); v2 += __shfl_xor(v2, off);
  }
  if (live && lane_a == 0) {
    double dl[3] = {0, 0, 0};
    if (var) {
      v0 += gp[3 * p]; v1 += gp[3 * p + 1]; v2 += gp[3 * p + 2];
      const double* Tp = T + 6 * p;
      dl[0] = -(Tp[0] * v0 + Tp[1] * v1 + Tp[2] * v2);
      dl[1] = -(Tp[1] * v0 + Tp[3] * v1 + Tp[4] * v2);
      dl[2] = -(Tp[2] * v0 + Tp[4] * v1 + Tp[5] * v2);
      double pp = 0.0;
#pragma unroll
      for (int j = 0; j < 3; ++j) pp += dl[j] * (Vdiag0[3 * p + j] * inv_radius * dl[j] - gp[3 * p + j]);
      if (limb) lpart.add(pp); else part += pp;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) delta_p[3 * p + j] = dl[j];
  }
  }
  if (limb) { limbs_block_add(lpart, limb + 1 * PXR_LIMBS, lsh); return; }
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(scal_sum + 1, red[0] + red[1] + red[2] + red[3]);
}

// ---- K_update: candidate = x (+) scale * delta --------------------------------------------------------
struct ParamPtrs { double* q; double* t; double* k; double* X; };
constexpr int UPDATE_IPT = 4;

__global__ __launch_bounds__(256) void k_update(const SolveDev d, const double* __restrict__ delta_c,
                                                const double* __restrict__ delta_p, ParamPtrs out,
                                                double* __restrict__ scal_rep, double* __restrict__ scal_sum, long long* __restrict__ limb) {
  // UPDATE_IPT items per thread and one pair of atomics per workgroup: the four norms are four addresses, and an atomic per
  // wavefront (3 100 of them at 200 000 points) serialised there for most of the kernel's 82 us
  __shared__ double red[4][256 / 64];
  __shared__ long long lsh[PXR_LIMBS * 4];
  const int n_img = d.v.n_images, n_cam = d.v.n_cameras;
  double step2_rep = 0, x2_rep = 0, step2_sum = 0, x2_sum = 0;
  Limbs l_step_rep, l_x_rep, l_step_sum, l_x_sum;          // deterministic mode: one limb addend per image / camera / point
  for (int rep = 0; rep < UPDATE_IPT; ++rep) {
  const int64_t tid = ((int64_t)blockIdx.x * UPDATE_IPT + rep) * blockDim.x + threadIdx.x;
  if (tid < n_img) {
    const int i = (int)tid;
    double q0[4], t0[3], q1[4], t1[3];
    for (int j = 0; j < 4; ++j) q1[j] = q0[j] = d.v.d_qvec[4 * i + j];
    for (int j = 0; j < 3; ++j) t1[j] = t0[j] = d.v.d_tvec[3 * i + j];
    if (d.pose_dim[i] > 0) {
      const int po = d.pose_off[i];
      const double dd[3] = {delta_c[po] * d.scale_c[po], delta_c[po + 1] * d.scale_c[po + 1],
                            delta_c[po + 2] * d.scale_c[po + 2]};
      const double nd = sqrt(dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
      if (nd != 0.0) {   // QuaternionManifold::Plus [upstream Ceres manifold.cc]
        const double sn = sin(nd) / nd;
        const double qd[4] = {cos(nd), sn * dd[0], sn * dd[1], sn * dd[2]};
        q1[0] = qd[0] * q0[0] - qd[1] * q0[1] - qd[2] * q0[2] - qd[3] * q0[3];
        q1[1] = qd[0] * q0[1] + qd[1] * q0[0] + qd[2] * q0[3] - qd[3] * q0[2];
        q1[2] = qd[0] * q0[2] - qd[1] * q0[3] + qd[2] * q0[0] + qd[3] * q0[1];
        q1[3] = qd[0] * q0[3] + qd[1] * q0[2] - qd[2] * q0[1] + qd[3] * q0[0];
      }
      int col = 3;
      for (int a = 0; a < 3; ++a) {
        if ((d.tmask[i] >> a) & 1) continue;
        t1[a] = t0[a] + delta_c[po + col] * d.scale_c[po + col];
        ++col;
      }
      double ix = 0.0, is = 0.0;
      for (int j = 0; j < 4; ++j) { ix += q0[j] * q0[j]; is += (q1[j] - q0[j]) * (q1[j] - q0[j]); }
      for (int j = 0; j < 3; ++j) { ix += t0[j] * t0[j]; is += (t1[j] - t0[j]) * (t1[j] - t0[j]); }
      if (limb) { l_x_rep.add(ix); l_step_rep.add(is); } else { x2_rep += ix; step2_rep += is; }
    }
    for (int j = 0; j < 4; ++j) out.q[4 * i + j] = q1[j];
    for (int j = 0; j < 3; ++j) out.t[3 * i + j] = t1[j];
  } else if (tid < n_img + n_cam) {
    const int c = (int)(tid - n_img);
    const int K = camera_num_params(d.v.d_cam_model[c]);
    int kc = 0;
    double ix = 0.0, is = 0.0;
    for (int a = 0; a < PXR_KPAD; ++a) {
      const double k0 = d.v.d_cam_params[(size_t)c * PXR_KPAD + a];
      double k1 = k0;
      if (d.intr_dim[c] > 0 && a < K) {
        if (!((d.cmask[c] >> a) & 1)) {
          const int io = d.intr_off[c] + kc;
          k1 = k0 + delta_c[io] * d.scale_c[io];
          ++kc;
        }
        ix += k0 * k0; is += (k1 - k0) * (k1 - k0);
      }
      out.k[(size_t)c * PXR_KPAD + a] = k1;
    }
    if (limb) { l_x_rep.add(ix); l_step_rep.add(is); } else { x2_rep += ix; step2_rep += is; }
  } else if (tid < n_img + n_cam + d.v.n_points) {
    const int64_t p = tid - n_img - n_cam;
    double ix = 0.0, is = 0.0;
    for (int j = 0; j < 3; ++j) {
      const double x0 = d.v.d_xyz[3 * p + j];
      double x1 = x0;
      if (d.pt_var[p]) {
        x1 = x0 + delta_p[3 * p + j] * d.scale_p[3 * p + j];
        ix += x0 * x0; is += (x1 - x0) * (x1 - x0);
      }
      out.X[3 * p + j] = x1;
    }
    if (limb) { l_x_sum.add(ix); l_step_sum.add(is); } else { x2_sum += ix; step2_sum += is; }
  }
  }
  if (limb) {
    limbs_block_add(l_step_rep, limb + (8 + 1) * PXR_LIMBS, lsh); __syncthreads();
    limbs_block_add(l_x_rep, limb + (8 + 2) * PXR_LIMBS, lsh); __syncthreads();
    limbs_block_add(l_step_sum, limb + 2 * PXR_LIMBS, lsh); __syncthreads();
    limbs_block_add(l_x_sum, limb + 3 * PXR_LIMBS, lsh);
    return;
  }
  step2_rep = wave_sum(step2_rep); x2_rep = wave_sum(x2_rep);
  step2_sum = wave_sum(step2_sum); x2_sum = wave_sum(x2_sum);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red[0][w] = step2_rep; red[1][w] = x2_rep; red[2][w] = step2_sum; red[3][w] = x2_sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r[4];
    for (int k = 0; k < 4; ++k) r[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
    if (r[0] != 0.0 || r[1] != 0.0) { atomicAdd(scal_rep + 1, r[0]); atomicAdd(scal_rep + 2, r[1]); }
    if (r[2] != 0.0 || r[3] != 0.0) { atomicAdd(scal_sum + 2, r[2]); atomicAdd(scal_sum + 3, r[3]); }
  }
}

__global__ void k_point_step_norm(int64_t n_points, const int* __restrict__ pt_var, const double* __restrict__ X0,
                                  const double* __restrict__ X1, double* __restrict__ out, long long* __restrict__ limb_slot) {
  __shared__ long long lsh[PXR_LIMBS * 4];
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0.0;
  if (p < n_points && pt_var[p])
    for (int j = 0; j < 3; ++j) { const double d = X1[3 * p + j] - X0[3 * p + j]; s += d * d; }
  if (limb_slot) { Limbs l; l.add(s); limbs_block_add(l, limb_slot, lsh); return; }   // deterministic mode
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0 && s != 0.0) atomicAdd(out, s);
}

__global__ void k_normalize_q(int n, double* __restrict__ q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double nn = sqrt(q[4 * i] * q[4 * i] + q[4 * i + 1] * q[4 * i + 1] + q[4 * i + 2] * q[4 * i + 2] + q[4 * i + 3] * q[4 * i + 3]);
  for (int j = 0; j < 4; ++j) q[4 * i + j] /= nn;   // colmap::Image::NormalizeQvec, bundle_optimizer.h:255
}

// gradient_tolerance test [upstream: max |g_i| <= tolerance, g in the unscaled variables]: the NUMBER of entries
// above the tolerance is accumulated instead of the maximum, because a count can be summed over the ranks with the
// all-reduce(sum) the solver already has (the point gradient is sharded; a rank-local maximum would let the ranks
// disagree about termination).
__global__ void k_count_above(int64_t n, const double* __restrict__ g, const double* __restrict__ scale, double tol,
                              double* __restrict__ out, long long* __restrict__ limb_slot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double v = (i < n && !(fabs(g[i] / scale[i]) <= tol)) ? 1.0 : 0.0;
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0 && v != 0.0) {
    if (limb_slot) atomicAdd(reinterpret_cast<unsigned long long*>(limb_slot + 1), (unsigned long long)((long long)v << 20));   // digit on the 2^-20 grid
    else atomicAdd(out, v);
  }
}

// Several ranks: only the upper triangle of [S | rhs] carries data (row r: columns r .. n, the last one the right-hand side --
// one contiguous run of n - r + 1 doubles), so the collective moves the packed n (n + 3) / 2 doubles instead of the n (n + 1)
// of the square (10.2 MB instead of 20.3 MB at 200 cameras).  One workgroup per row, coalesced both ways.
__device__ __forceinline__ int64_t packed_row_offset(int n, int r) { return (int64_t)r * (n + 1) - (int64_t)r * (r - 1) / 2; }
__global__ __launch_bounds__(256) void k_pack_upper(int n, const double* __restrict__ S, double* __restrict__ packed) {
  const int r = blockIdx.x, len = n - r + 1;
  const double* src = S + (size_t)r * (n + 1) + r;
  double* dst = packed + packed_row_offset(n, r);
  for (int x = threadIdx.x; x < len; x += blockDim.x) dst[x] = src[x];
}
__global__ __launch_bounds__(256) void k_unpack_upper(int n, const double* __restrict__ packed, double* __restrict__ S) {
  const int r = blockIdx.x, len = n - r + 1;
  double* dst = S + (size_t)r * (n + 1) + r;
  const double* src = packed + packed_row_offset(n, r);
  for (int x = threadIdx.x; x < len; x += blockDim.x) dst[x] = src[x];
}

// cost = sum 0.5 rho(s) over the records as limbs (pxr_device.h): one addend per observation, integer atomics per workgroup --
// the same integers whatever the launch shape or the rank count.  (The cost fused into the residual kernel was one
// floating-point atomic per wavefront on one address: 50 us of serialisation at 1M observations.)
__global__ __launch_bounds__(256) void k_cost_limbs(const double* __restrict__ rec, int64_t n, pxr_loss loss, long long* __restrict__ slot) {
  __shared__ long long lsh[PXR_LIMBS * 4];
  Limbs l;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double rho[3];
    loss_eval(loss.type, loss.a, 1.0, rec[i * PXR_OBS_REC], rho);
    l.add(0.5 * rho[0]);
  }
  limbs_block_add(l, slot, lsh);
}
// x[0 .. n) added to a limb slot, one addend per entry (the inner iterations' per-point costs)
__global__ __launch_bounds__(256) void k_limb_accumulate(const double* __restrict__ x, int64_t n, long long* __restrict__ slot) {
  __shared__ long long lsh[PXR_LIMBS * 4];
  Limbs l;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) l.add(x[i]);
  limbs_block_add(l, slot, lsh);
}
// scal[i] = value of limb slot i for every i in `mask` (16 slots)
__global__ void k_limbs_finish(const long long* __restrict__ limb, unsigned mask, double* __restrict__ scal) {
  const int i = threadIdx.x;
  if (i < 16 && ((mask >> i) & 1u)) scal[i] = limb_value(limb + i * PXR_LIMBS);
}
// deterministic linearisation, before the ranks' sum: diag(U) (raw fixed-point slots) next to g_c, and the chunks' trace
// contributions as limbs behind them -- ONE workgroup (n_c is a few thousand at most)
__global__ __launch_bounds__(1024) void k_diag_and_trace(int n, const double* __restrict__ U, double* __restrict__ diag,
                                                          const double* __restrict__ chunk_trace, int n_chunks, long long* __restrict__ trace_out) {
  __shared__ double sh[1024];
  for (int i = threadIdx.x; i < n; i += 1024) diag[i] = U[(size_t)i * n + i];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_chunks; i += 1024) acc += chunk_trace[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    Limbs l; l.add(sh[0]);
    for (int k = 0; k < PXR_LIMBS; ++k) trace_out[k] = l.q[k];
  }
}
// ... and after it: the slots of diag(U) | g_c become doubles, and the overflow guard's statistics of the diagonal
__global__ __launch_bounds__(1024) void k_finish_and_stats(int n, int n_slots, double* __restrict__ gcd, double det_scale,
                                                            const long long* __restrict__ trace_limbs, double* __restrict__ out4) {
  __shared__ double smax[1024], ssum[1024], smin[1024];
  double mx = 0.0, sm = 0.0, mn = 0.0;
  for (int i = threadIdx.x; i < n_slots; i += 1024) {
    const double v = accum_value(gcd[i], det_scale);
    gcd[i] = v;
    if (i < n) { mx = fmax(mx, v); mn = fmin(mn, v); sm += v; }
  }
  smax[threadIdx.x] = mx; ssum[threadIdx.x] = sm; smin[threadIdx.x] = mn;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + off]);
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + off]);
      ssum[threadIdx.x] += ssum[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out4[0] = smax[0]; out4[1] = ssum[0]; out4[2] = smin[0]; out4[3] = limb_value(trace_limbs); }
}
// 64-bit integers through a sum-of-doubles collective (the caller's all-reduce callback): each integer as its signed high and
// unsigned low 32-bit halves -- sums of up to 2^20 of them are exact in a double -- and back (modulo 2^64, like the integers)
__global__ void k_i64_split(int64_t n, const long long* __restrict__ in, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long v = in[i];
  out[2 * i] = (double)(int)(v >> 32);
  out[2 * i + 1] = (double)(unsigned)(v & 0xffffffffll);
}
__global__ void k_i64_join(int64_t n, const double* __restrict__ in, long long* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long hi = (unsigned long long)(long long)in[2 * i], lo = (unsigned long long)(long long)in[2 * i + 1];
  out[i] = (long long)((hi << 32) + lo);
}

// keep rank 0's copy of a replicated buffer: the other ranks zero theirs before an all-reduce(sum)
__global__ void k_zero(int64_t n, double* __restrict__ x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = 0.0;
}

// ---- set-up: flattened index chains of the Schur / back-substitution kernels ------------------------------------
// obs_cols[o] / part_obs[o]: column descriptor and observation id of slot o of the point-ordered list;
// so[o]: {observation, point, first partner slot, partner count (0 = constant point)} of slot o of the image-ordered list
__global__ __launch_bounds__(256) void k_build_descriptors(int64_t n_obs, const int32_t* __restrict__ obs_image,
                                                           const int32_t* __restrict__ obs_point,
                                                           const int32_t* __restrict__ image_camera,
                                                           const int64_t* __restrict__ pt_obs, const int64_t* __restrict__ img_obs,
                                                           const int64_t* __restrict__ pt_ptr, const int* __restrict__ pt_var,
                                                           const int* __restrict__ pose_off, const int* __restrict__ pose_dim,
                                                           const int* __restrict__ intr_off, const int* __restrict__ intr_dim,
                                                           int4* __restrict__ obs_cols, int* __restrict__ part_obs,
                                                           int4* __restrict__ so) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= n_obs) return;
  const int64_t j = pt_obs[o];
  const int im = obs_image[j], cm = image_camera[im];
  obs_cols[o] = make_int4(pose_off[im], pose_dim[im], intr_off[cm], intr_dim[cm]);
  part_obs[o] = (int)j;
  const int64_t i = img_obs[o];
  const int64_t pt = obs_point[i];
  so[o] = make_int4((int)i, (int)pt, (int)pt_ptr[pt], pt_var[pt] ? (int)(pt_ptr[pt + 1] - pt_ptr[pt]) : 0);
}

// ---- observation lists on the device (set-up) -------------------------------------------------------------------------
// counts per image / per point, index range check, and whether the observations are ordered by point (then the
// point-ordered list is the identity); flags[0] = out-of-range index seen, flags[1] = a point index decreases
// (the image counts are histogrammed in LDS per chunk of COUNT_CHUNK observations: a million global atomics on the ~13 cache
//  lines of 200 image counters took 0.5 ms of every solve, profiles/r6_lm_setup.txt)
constexpr int COUNT_CHUNK = 4096;
__global__ __launch_bounds__(256) void k_count_indices(int64_t n_obs, const int32_t* __restrict__ obs_image,
                                                       const int32_t* __restrict__ obs_point, int n_img, int64_t n_pts,
                                                       unsigned long long* __restrict__ img_cnt, unsigned long long* __restrict__ pt_cnt,
                                                       int* __restrict__ flags) {
  extern __shared__ unsigned int sh_img[];                 // n_img counters (n_img <= SORT_MAX_IMAGES)
  for (int k = threadIdx.x; k < n_img; k += 256) sh_img[k] = 0u;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * COUNT_CHUNK, i1 = min(n_obs, i0 + COUNT_CHUNK);
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const int im = obs_image[i], pt = obs_point[i];
    if (im < 0 || im >= n_img || pt < 0 || pt >= n_pts) { atomicOr(&flags[0], 1); continue; }
    atomicAdd(&sh_img[im], 1u);
    atomicAdd(&pt_cnt[pt + 1], 1ull);
    if (i > 0 && obs_point[i - 1] > pt) atomicOr(&flags[1], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_img; k += 256)
    if (sh_img[k]) atomicAdd(&img_cnt[k + 1], (unsigned long long)sh_img[k]);
}
// the per-point side of the structure without a trip to the host (200k points: 1.6 MB each way through pageable memory were
// 0.5 ms of every solve): pt_ptr = prefix sums of the counts k_count_indices left in cnt[p + 1] (cnt[0] = 0), pt_var[p] = the point
// is not constant and has an observation, *n_var = how many.  Two launches: the sums of chunks of PT_SCAN_CHUNK points, then
// every workgroup adds up the chunks before its own and scans its chunk.
constexpr int PT_SCAN_CHUNK = 2048;
__device__ __forceinline__ unsigned long long block_sum_256(unsigned long long v, unsigned long long* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) sh[t] += sh[t + o]; __syncthreads(); }
  const unsigned long long r = sh[0];
  __syncthreads();
  return r;
}
__global__ __launch_bounds__(256) void k_pt_scan_partials(int64_t n_pts, const unsigned long long* __restrict__ cnt, unsigned long long* __restrict__ part) {
  __shared__ unsigned long long sh[256];
  const int64_t p0 = (int64_t)blockIdx.x * PT_SCAN_CHUNK;
  unsigned long long s = 0;
  for (int j = threadIdx.x; j < PT_SCAN_CHUNK; j += 256) if (p0 + j < n_pts) s += cnt[p0 + j + 1];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pt_scan_apply(int64_t n_pts, const unsigned long long* __restrict__ cnt, const unsigned long long* __restrict__ part,
                                                       const uint8_t* __restrict__ pt_const, int64_t* __restrict__ pt_ptr, int* __restrict__ pt_var,
                                                       unsigned long long* __restrict__ n_var) {
  __shared__ unsigned long long sh[256];
  constexpr int PER = PT_SCAN_CHUNK / 256;
  const int t = threadIdx.x;
  unsigned long long base = 0;
  for (int g = t; g < (int)blockIdx.x; g += 256) base += part[g];
  base = block_sum_256(base, sh);
  const int64_t p0 = (int64_t)blockIdx.x * PT_SCAN_CHUNK + (int64_t)t * PER;
  unsigned long long c[PER], mine = 0;
  int nv = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    c[j] = p0 + j < n_pts ? cnt[p0 + j + 1] : 0ull;
    mine += c[j];
  }
  sh[t] = mine;                                             // inclusive scan of the threads' sums (Hillis-Steele, 8 rounds)
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned long long add = t >= o ? sh[t - o] : 0ull;
    __syncthreads();
    sh[t] += add;
    __syncthreads();
  }
  unsigned long long run = base + sh[t] - mine;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (p0 + j >= n_pts) break;
    run += c[j];
    pt_ptr[p0 + j + 1] = (int64_t)run;
    const int v = (!pt_const[p0 + j] && c[j] > 0) ? 1 : 0;
    pt_var[p0 + j] = v; nv += v;
  }
  if (blockIdx.x == 0 && t == 0) pt_ptr[0] = 0;
  __syncthreads();
  const unsigned long long tot = block_sum_256((unsigned long long)nv, sh);
  if (t == 0 && tot) atomicAdd(n_var, tot);
}
__global__ void k_iota(int64_t n, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i;
}
// STABLE counting sort of the observation ids by image (the order the host fill produces: ascending observation id inside
// an image -- the summation order of k_img and of the Schur contraction, which must not depend on the run):
//   k_sort_hist     a workgroup counts the images of its contiguous chunk of SORT_CHUNK observations,
//   k_sort_offsets  per image, an exclusive scan of those counts over the chunks on top of the image's first slot,
//   k_sort_scatter  the workgroup walks its chunk in order, 256 observations a round: slot = running offset of the image +
//                   number of EARLIER observations of the round with the same image.
constexpr int SORT_CHUNK = 4096, SORT_MAX_IMAGES = 8192;
__global__ __launch_bounds__(256) void k_sort_hist(int64_t n_obs, const int32_t* __restrict__ obs_image, int n_img,
                                                   int* __restrict__ hist) {
  extern __shared__ int sh_cnt[];
  for (int k = threadIdx.x; k < n_img; k += blockDim.x) sh_cnt[k] = 0;
  __syncthreads();
  const int64_t b0 = (int64_t)blockIdx.x * SORT_CHUNK, b1 = min(n_obs, b0 + SORT_CHUNK);
  for (int64_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) atomicAdd(&sh_cnt[obs_image[i]], 1);
  __syncthreads();
  for (int k = threadIdx.x; k < n_img; k += blockDim.x) hist[(size_t)blockIdx.x * n_img + k] = sh_cnt[k];
}
__global__ void k_sort_offsets(int n_img, int n_chunks, const int64_t* __restrict__ img_ptr, int* __restrict__ hist) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_img) return;
  int run = (int)img_ptr[k];
  for (int b = 0; b < n_chunks; ++b) {
    const int c = hist[(size_t)b * n_img + k];
    hist[(size_t)b * n_img + k] = run;
    run += c;
  }
}
__global__ __launch_bounds__(256) void k_sort_scatter(int64_t n_obs, const int32_t* __restrict__ obs_image, int n_img,
                                                      const int* __restrict__ hist, int64_t* __restrict__ img_obs) {
  extern __shared__ int sh_off[];          // [n_img] running offsets, then [256] the keys of the round
  int* keys = sh_off + n_img;
  for (int k = threadIdx.x; k < n_img; k += blockDim.x) sh_off[k] = hist[(size_t)blockIdx.x * n_img + k];
  const int64_t b0 = (int64_t)blockIdx.x * SORT_CHUNK, b1 = min(n_obs, b0 + SORT_CHUNK);
  for (int64_t r0 = b0; r0 < b1; r0 += 256) {
    const int64_t i = r0 + threadIdx.x;
    const int key = i < b1 ? obs_image[i] : -1;
    __syncthreads();                       // offsets initialised / updated by the previous round
    keys[threadIdx.x] = key;
    __syncthreads();
    int before = 0, after = 0;
    if (key >= 0) {
      for (int t = 0; t < 256; ++t) {
        const int same = keys[t] == key;
        before += same & (t < (int)threadIdx.x);
        after += same & (t > (int)threadIdx.x);
      }
      img_obs[sh_off[key] + before] = i;
    }
    __syncthreads();                       // every slot of the round is computed from the old offsets
    if (key >= 0 && after == 0) sh_off[key] += before + 1;
  }
}

// ---- host orchestration -----------------------------------------------------------------------------------
// The ~60 work buffers of a solve come out of ONE grow-only allocation of the context (round 6): the first solve of a size takes
// them from hipMalloc and records what it needed, the next ones bump a pointer -- ~60 hipMalloc at the start and ~60 hipFree at the
// end (each a device synchronisation) were 4 ms of a 24 ms call at 1M observations (profiles/r6_lm_setup.txt).  The arena of the
// solve that is running on this thread; NULL outside pxr_ba_solve and with PXR_BA_ARENA=0.
struct SolveArena { char* base = nullptr; size_t cap = 0, used = 0, wanted = 0; };
static thread_local SolveArena* g_solve_arena = nullptr;
void* solve_scratch(size_t bytes, bool* owned) {
  const size_t b = ((bytes ? bytes : 1) + 255) & ~(size_t)255;
  if (SolveArena* a = g_solve_arena) {
    a->wanted += b;
    if (a->used + b <= a->cap) { void* p = a->base + a->used; a->used += b; *owned = false; return p; }
  }
  void* p = nullptr;
  *owned = true;
  if (hipMalloc(&p, b) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool owned = true;
  int alloc(size_t count) {
    n = count;
    p = static_cast<T*>(solve_scratch(sizeof(T) * (count ? count : 1), &owned));
    return p ? PXR_OK : set_error(PXR_ENOMEM, "hipMalloc(solver buffer): %zu bytes", sizeof(T) * count);
  }
  int upload(const std::vector<T>& h, hipStream_t s) {
    int rc = alloc(h.size());
    if (rc) return rc;
    if (!h.empty()) return hip_check(hipMemcpyAsync(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, s), "H2D");
    return PXR_OK;
  }
  ~DevBuf() { if (p && owned) (void)hipFree(p); }
};

// pxr_chol.hip: factor the n x n SPD system stored row-major (upper) in the n x (n + 1) buffer `a` whose last
// column is the right-hand side (forward substitution is fused into the factorisation), then back-substitute.
int chol_factor_solve(hipStream_t st, double* a, int n, int* d_info, double* linv_ws, double* x_out, bool zero_info = true);
size_t chol_workspace_doubles(int n);
size_t inner_wave_stage_bytes(int64_t n_pts);        // pxr_ba_inner.hip: host room for the packed table of n_pts points, worst case

// pxr_ba_inner.hip
int launch_inner_iterations(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                            const pxr_loss* loss, const int64_t* d_pt_ptr, const int64_t* d_pt_obs,
                            const int* d_pt_var, double* d_cost_before, const InnerLists* lists, double* d_cost_per_point,
                            const GramCache* gram, bool gram_warm);
int make_inner_lists(pxr_ctx* ctx, const int64_t* h_pt_ptr, int64_t n_pts, void* h_wave_stage, size_t wave_stage_bytes, const pxr_ba_view* view, const int64_t* d_pt_ptr, const int64_t* d_pt_obs,
                     InnerLists* out);
void free_inner_lists(InnerLists* l);

static inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)((n + b - 1) / b); }

#define RC(call) do { int _rc = (call); if (_rc != PXR_OK) return _rc; } while (0)
#define LAUNCH_CHECK(name) RC(hip_check(hipGetLastError(), name))

}  // namespace pxr

extern "C" int pxr_ba_solve(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                            const pxr_loss* loss, const uint8_t* h_pose_const, const uint8_t* h_tvec_const_mask,
                            const uint16_t* h_cam_const_mask, const uint8_t* h_point_const,
                            const pxr_lm_options* opt, pxr_allreduce_fn allreduce, void* ar_user,
                            pxr_lm_summary* sum) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && view && cfg && loss && opt && sum, "pxr_ba_solve: NULL argument");
  PXR_REQUIRE(h_pose_const && h_tvec_const_mask && h_cam_const_mask && h_point_const,
              "pxr_ba_solve: NULL parameterisation array");
  PXR_REQUIRE(view->n_obs > 0, "pxr_ba_solve: problem has no residuals (bundle_optimizer.h:174-176)");
  PXR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const auto t_setup0 = std::chrono::steady_clock::now();
  // (declared before every buffer: destroyed after all of them -- then the arena grows to what this solve asked for)
  struct ArenaScope {
    pxr_ctx* ctx; SolveArena a; bool on;
    explicit ArenaScope(pxr_ctx* c) : ctx(c) {
      const char* e = std::getenv("PXR_BA_ARENA");
      on = !(e && e[0] == '0');
      a.base = static_cast<char*>(c->d_solve_arena); a.cap = c->solve_arena_bytes;
      if (on) g_solve_arena = &a;
    }
    ~ArenaScope() {
      g_solve_arena = nullptr;
      if (!on || a.wanted <= a.cap) return;
      (void)hipDeviceSynchronize();                         // (an error return may leave kernels of this solve in flight)
      if (ctx->d_solve_arena) (void)hipFree(ctx->d_solve_arena);
      ctx->d_solve_arena = nullptr; ctx->solve_arena_bytes = 0;
      const size_t want = a.wanted + a.wanted / 16 + (1u << 20);
      if (hipMalloc(&ctx->d_solve_arena, want) == hipSuccess) ctx->solve_arena_bytes = want;
      else { (void)hipGetLastError(); ctx->d_solve_arena = nullptr; }     // no room to keep it: the next solve allocates per buffer again
    }
  } arena_scope(ctx);
  const bool setup_verbose = std::getenv("PXR_VERBOSE") != nullptr;
  auto setup_mark = [&](const char* what) {
    if (setup_verbose)
      fprintf(stderr, "[pxr_ba_solve] setup: %-28s at %.2f ms\n", what,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_setup0).count());
  };
  const int64_t n_obs = view->n_obs, n_pts = view->n_points;
  const int n_img = view->n_images, n_cam = view->n_cameras;

  // ---- structure: which blocks are in the program, offsets, observation lists per image and per point -----------------
  // Fast path (observations ordered by point -- what every caller of this library produces -- and at most
  // SORT_MAX_IMAGES images): the 2 x n_obs index arrays never leave the device; only the per-image / per-point COUNTS come
  // to the host, the lists are built by a stable counting sort on the device.  Otherwise (and with PXR_BA_SETUP_HOST=1,
  // for the equivalence test) the lists are built on the host from a copy of the index arrays.
  std::vector<int32_t> obs_image, obs_point, image_camera(n_img), cam_model(n_cam);
  std::vector<int64_t> img_cnt(n_img + 1, 0), pt_cnt;      // (pt_cnt: the general host path only -- the fast path keeps the per-point side on the device)
  DevBuf<unsigned long long> d_cnt;
  DevBuf<int> d_flags;
  bool device_lists = n_img <= SORT_MAX_IMAGES && std::getenv("PXR_BA_SETUP_HOST") == nullptr;
  PXR_HIP(hipMemcpyAsync(image_camera.data(), view->d_image_camera, 4 * n_img, hipMemcpyDeviceToHost, st));
  PXR_HIP(hipMemcpyAsync(cam_model.data(), view->d_cam_model, 4 * n_cam, hipMemcpyDeviceToHost, st));
  setup_mark("camera tables read");
  if (device_lists) {
    int h_flags[2] = {0, 0};
    RC(d_cnt.alloc((size_t)n_img + 1 + (size_t)n_pts + 1 + 1)); RC(d_flags.alloc(2));     // ... + the number of variable points
    setup_mark("counter buffers");
    PXR_HIP(hipMemsetAsync(d_cnt.p, 0, sizeof(unsigned long long) * d_cnt.n, st));
    PXR_HIP(hipMemsetAsync(d_flags.p, 0, sizeof(int) * 2, st));
    hipLaunchKernelGGL(k_count_indices, dim3((unsigned)((n_obs + COUNT_CHUNK - 1) / COUNT_CHUNK)), dim3(256), sizeof(unsigned int) * n_img, st, n_obs, view->d_obs_image,
                       view->d_obs_point, n_img, n_pts, d_cnt.p, d_cnt.p + n_img + 1, d_flags.p);
    static_assert(sizeof(unsigned long long) == sizeof(int64_t), "counter width");
    PXR_HIP(hipMemcpyAsync(img_cnt.data(), d_cnt.p, sizeof(int64_t) * (n_img + 1), hipMemcpyDeviceToHost, st));
    PXR_HIP(hipMemcpyAsync(h_flags, d_flags.p, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
    PXR_HIP(hipStreamSynchronize(st));
    PXR_REQUIRE(h_flags[0] == 0, "pxr_ba_solve: an observation references an image / point out of range");
    if (h_flags[1]) {                       // not ordered by point: the general host path
      device_lists = false;
      std::fill(img_cnt.begin(), img_cnt.end(), 0);
    }
  }
  if (!device_lists) {
    pt_cnt.assign(n_pts + 1, 0);
    obs_image.resize(n_obs); obs_point.resize(n_obs);
    PXR_HIP(hipMemcpyAsync(obs_image.data(), view->d_obs_image, 4 * n_obs, hipMemcpyDeviceToHost, st));
    PXR_HIP(hipMemcpyAsync(obs_point.data(), view->d_obs_point, 4 * n_obs, hipMemcpyDeviceToHost, st));
  }
  PXR_HIP(hipStreamSynchronize(st));
  setup_mark(device_lists ? "counts on the host" : "index arrays on the host");
  static const int kNumParams[11] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};   // [upstream COLMAP 3.8] kNumParams by model id
  for (int c = 0; c < n_cam; ++c)
    PXR_REQUIRE(cam_model[c] >= 0 && cam_model[c] <= 10, "pxr_ba_solve: unsupported camera model id %d", cam_model[c]);
  if (!device_lists) {
    for (int64_t i = 0; i < n_obs; ++i) {
      PXR_REQUIRE(obs_image[i] >= 0 && obs_image[i] < n_img && obs_point[i] >= 0 && obs_point[i] < n_pts,
                  "pxr_ba_solve: observation %lld references image %d / point %d out of range", (long long)i,
                  obs_image[i], obs_point[i]);
      ++img_cnt[obs_image[i] + 1]; ++pt_cnt[obs_point[i] + 1];
    }
  }
  // With several ranks a camera-side block may have no local observation but still be part of the
  // (global) program: the caller marks unused blocks constant, we keep every non-constant block.
  std::vector<int> pose_off(n_img), pose_dim(n_img), tmask(n_img), intr_off(n_cam), intr_dim(n_cam), cmask(n_cam), pt_var;
  int off = 0, dpose_max = 0, dintr_max = 0;
  for (int i = 0; i < n_img; ++i) {
    int d = 0;
    tmask[i] = h_tvec_const_mask[i] & 7;
    if (!h_pose_const[i]) d = 3 + (3 - __builtin_popcount(tmask[i]));
    pose_off[i] = off; pose_dim[i] = d; off += d;
    dpose_max = std::max(dpose_max, d);
  }
  for (int c = 0; c < n_cam; ++c) {
    const int K = kNumParams[cam_model[c]];
    cmask[c] = h_cam_const_mask[c] & ((1 << K) - 1);
    const int d = K - __builtin_popcount(cmask[c]);
    intr_off[c] = off; intr_dim[c] = d; off += d;
    dintr_max = std::max(dintr_max, d);
  }
  const int n_c = off;
  const int DC = std::max(1, dpose_max + dintr_max);
  const int LS = 11 + 2 * DC;
  // linear solver by image count, like bundle_optimizer.h:180-191 overrides the user's choice
  PXR_REQUIRE(opt->linear_solver >= PXR_LINEAR_AUTO && opt->linear_solver <= PXR_LINEAR_ITERATIVE,
              "pxr_ba_solve: unknown linear_solver %d", opt->linear_solver);
  const bool iterative = n_c > 0 && (opt->linear_solver == PXR_LINEAR_ITERATIVE ||
                                     (opt->linear_solver == PXR_LINEAR_AUTO && n_img > PXR_MAX_IMAGES_DIRECT));
  sum->linear_solver = iterative ? PXR_LINEAR_ITERATIVE : PXR_LINEAR_DIRECT;
  sum->linear_iterations = 0; sum->collective_kib = 0;
  int64_t n_pvar = 0;
  if (!device_lists) {
    pt_var.resize(n_pts);
    for (int64_t p = 0; p < n_pts; ++p) { pt_var[p] = (!h_point_const[p] && pt_cnt[p + 1] > 0) ? 1 : 0; n_pvar += pt_var[p]; }
    PXR_REQUIRE(n_c > 0 || n_pvar > 0, "pxr_ba_solve: every parameter block is constant");
    for (int64_t p = 0; p < n_pts; ++p) pt_cnt[p + 1] += pt_cnt[p];
  }
  for (int i = 0; i < n_img; ++i) img_cnt[i + 1] += img_cnt[i];
  std::vector<int64_t> img_obs, pt_obs;
  if (!device_lists) {
    img_obs.resize(n_obs); pt_obs.resize(n_obs);
    std::vector<int64_t> ic(img_cnt.begin(), img_cnt.end() - 1), pc(pt_cnt.begin(), pt_cnt.end() - 1);
    for (int64_t i = 0; i < n_obs; ++i) { img_obs[ic[obs_image[i]]++] = i; pt_obs[pc[obs_point[i]]++] = i; }
  }
  setup_mark(device_lists ? "host prefix sums" : "host CSR (counts, fill)");
  std::vector<ImgChunk> chunks;
  const int64_t CH = 512;    // observations per k_img workgroup (4 LDS batches)
  for (int i = 0; i < n_img; ++i)
    for (int64_t b = img_cnt[i]; b < img_cnt[i + 1]; b += CH)
      chunks.push_back({i, b, std::min(img_cnt[i + 1], b + CH)});

  // Schur contraction: larger chunks (fewer LDS flushes), and per-observation column descriptors so the
  // inner loop does not chase obs -> image -> camera -> offsets
  std::vector<ImgChunk> schur_chunks;
  const int64_t SCH = 1024;
  for (int i = 0; i < n_img; ++i)
    for (int64_t b = img_cnt[i]; b < img_cnt[i + 1]; b += SCH)
      schur_chunks.push_back({i, b, std::min(img_cnt[i + 1], b + SCH)});
  PXR_REQUIRE(n_obs < ((int64_t)1 << 31), "pxr_ba_solve: more than 2^31 observations per rank");
  // (the per-observation descriptors -- partner column descriptors in pt_obs order, slots in img_obs order -- are
  // filled on the device by k_build_descriptors from the arrays uploaded below)
  // preconditioner blocks of the iterative solver: the pose columns of an image and the intrinsics columns of a
  // camera; one joint block where the camera belongs to a single image (every column in exactly one block)
  std::vector<int2> col_group;
  std::vector<int> group_size, group_cols;
  if (iterative) {
    PXR_REQUIRE(DC <= PCG_GS, "pxr_ba_solve: %d camera-side columns per observation exceed the preconditioner block size", DC);
    std::vector<int> cam_users(n_cam, 0);
    for (int i = 0; i < n_img; ++i) ++cam_users[image_camera[i]];
    col_group.assign(n_c, make_int2(-1, -1));
    auto open_group = [&]() { group_size.push_back(0); group_cols.resize(group_cols.size() + PCG_GS, 0); return (int)group_size.size() - 1; };
    auto add_cols = [&](int g, int first, int count) {
      for (int a = 0; a < count; ++a) {
        col_group[first + a] = make_int2(g, group_size[g]);
        group_cols[(size_t)g * PCG_GS + group_size[g]++] = first + a;
      }
    };
    for (int i = 0; i < n_img; ++i) {
      const int c = image_camera[i];
      const bool joint = cam_users[c] == 1 && intr_dim[c] > 0;
      if (pose_dim[i] == 0 && !joint) continue;
      const int g = open_group();
      add_cols(g, pose_off[i], pose_dim[i]);
      if (joint) add_cols(g, intr_off[c], intr_dim[c]);
    }
    for (int c = 0; c < n_cam; ++c)
      if (cam_users[c] != 1 && intr_dim[c] > 0) add_cols(open_group(), intr_off[c], intr_dim[c]);
    for (int c = 0; c < n_c; ++c) PXR_REQUIRE(col_group[c].x >= 0, "pxr_ba_solve: internal: column %d in no preconditioner block", c);
  }
  setup_mark("host structure (CSR, chunks)");
  // ---- device buffers ---------------------------------------------------------------------------
  DevBuf<int> d_pose_off, d_pose_dim, d_tmask, d_intr_off, d_intr_dim, d_cmask, d_pt_var;
  DevBuf<int64_t> d_img_obs, d_pt_ptr, d_pt_obs;
  DevBuf<ImgChunk> d_chunks, d_schur_chunks;
  DevBuf<int4> d_obs_cols, d_so;
  DevBuf<int> d_part_obs;
  RC(d_pose_off.upload(pose_off, st)); RC(d_pose_dim.upload(pose_dim, st)); RC(d_tmask.upload(tmask, st));
  RC(d_intr_off.upload(intr_off, st)); RC(d_intr_dim.upload(intr_dim, st)); RC(d_cmask.upload(cmask, st));
  RC(d_chunks.upload(chunks, st));
  DevBuf<uint8_t> d_pt_const;
  DevBuf<unsigned long long> d_pt_part;
  unsigned long long h_n_pvar = 0;
  if (!device_lists) { RC(d_pt_var.upload(pt_var, st)); RC(d_pt_ptr.upload(pt_cnt, st)); }
  if (device_lists) {
    // offsets, variable flags and their number from the counts that never left the device (k_pt_scan_*)
    const int n_parts = (int)((n_pts + PT_SCAN_CHUNK - 1) / PT_SCAN_CHUNK);
    unsigned long long* const d_pt_cnt = d_cnt.p + n_img + 1;
    RC(d_pt_var.alloc(n_pts)); RC(d_pt_ptr.alloc((size_t)n_pts + 1)); RC(d_pt_const.alloc(n_pts)); RC(d_pt_part.alloc(n_parts));
    PXR_HIP(hipMemcpyAsync(d_pt_const.p, h_point_const, (size_t)n_pts, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pt_scan_partials, dim3(n_parts), dim3(256), 0, st, n_pts, (const unsigned long long*)d_pt_cnt, d_pt_part.p);
    hipLaunchKernelGGL(k_pt_scan_apply, dim3(n_parts), dim3(256), 0, st, n_pts, (const unsigned long long*)d_pt_cnt, (const unsigned long long*)d_pt_part.p,
                       (const uint8_t*)d_pt_const.p, d_pt_ptr.p, d_pt_var.p, d_pt_cnt + n_pts + 1);
    PXR_HIP(hipMemcpyAsync(&h_n_pvar, d_pt_cnt + n_pts + 1, sizeof(h_n_pvar), hipMemcpyDeviceToHost, st));
    RC(d_img_obs.alloc(n_obs)); RC(d_pt_obs.alloc(n_obs));
    hipLaunchKernelGGL(k_iota, dim3(nblk(n_obs)), dim3(256), 0, st, n_obs, d_pt_obs.p);     // ordered by point already
    DevBuf<int64_t> d_img_ptr;
    DevBuf<int> d_hist;
    const int n_chunks = (int)((n_obs + SORT_CHUNK - 1) / SORT_CHUNK);
    RC(d_img_ptr.upload(img_cnt, st)); RC(d_hist.alloc((size_t)n_chunks * n_img));
    hipLaunchKernelGGL(k_sort_hist, dim3(n_chunks), dim3(256), sizeof(int) * n_img, st, n_obs, view->d_obs_image, n_img, d_hist.p);
    hipLaunchKernelGGL(k_sort_offsets, dim3((n_img + 255) / 256), dim3(256), 0, st, n_img, n_chunks, d_img_ptr.p, d_hist.p);
    hipLaunchKernelGGL(k_sort_scatter, dim3(n_chunks), dim3(256), sizeof(int) * (n_img + 256), st, n_obs, view->d_obs_image, n_img,
                       d_hist.p, d_img_obs.p);
    LAUNCH_CHECK("observation-list kernels");
    setup_mark("list kernels launched");
    PXR_HIP(hipStreamSynchronize(st));      // d_img_ptr / d_hist go out of scope
    setup_mark("list kernels done");
    n_pvar = (int64_t)h_n_pvar;
    PXR_REQUIRE(n_c > 0 || n_pvar > 0, "pxr_ba_solve: every parameter block is constant");
  } else {
    RC(d_img_obs.upload(img_obs, st)); RC(d_pt_obs.upload(pt_obs, st));
  }
  RC(d_schur_chunks.upload(schur_chunks, st));
  RC(d_obs_cols.alloc(n_obs)); RC(d_so.alloc(n_obs)); RC(d_part_obs.alloc(n_obs));
  hipLaunchKernelGGL(k_build_descriptors, dim3(nblk(n_obs)), dim3(256), 0, st, n_obs, view->d_obs_image, view->d_obs_point,
                     view->d_image_camera, d_pt_obs.p, d_img_obs.p, d_pt_ptr.p, d_pt_var.p, d_pose_off.p, d_pose_dim.p,
                     d_intr_off.p, d_intr_dim.p, d_obs_cols.p, d_part_obs.p, d_so.p);
  LAUNCH_CHECK("k_build_descriptors");
  setup_mark("index uploads");
  const size_t nc1 = n_c ? n_c : 1;
  DevBuf<double> L, V, gp, Vd0, T, W, U, S /* S | rhs */, gcd /* diagU | g_c */, damp_c, scale_c, scale_p,
      delta_c, delta_p, rec_a, rec_b, q1, t1, k1, X1, scal;
  RC(L.alloc((size_t)n_obs * LS)); RC(V.alloc((size_t)n_pts * 6)); RC(gp.alloc((size_t)n_pts * 3));
  RC(Vd0.alloc((size_t)n_pts * 3)); RC(T.alloc((size_t)n_pts * 6));
  RC(W.alloc((size_t)n_obs * DC * 3));
  // S: n_c x (n_c + 1) [S | rhs] + one spare row for the factorisation
  // direct solver: dense U and [S | rhs]; iterative solver: one DC x DC block of U per image, nothing quadratic in n_c
  RC(U.alloc(iterative ? (size_t)n_img * DC * DC : nc1 * nc1)); RC(S.alloc(iterative ? 1 : (nc1 + 1) * (nc1 + 1)));
  RC(gcd.alloc(2 * nc1 + 8)); RC(damp_c.alloc(nc1));   // diag(U) | g_c | trace limbs of the overflow guard
  RC(scale_c.alloc(nc1)); RC(scale_p.alloc((size_t)n_pts * 3)); RC(delta_c.alloc(nc1)); RC(delta_p.alloc((size_t)n_pts * 3));
  RC(rec_a.alloc((size_t)n_obs * PXR_OBS_REC)); RC(rec_b.alloc((size_t)n_obs * PXR_OBS_REC));
  RC(q1.alloc((size_t)n_img * 4)); RC(t1.alloc((size_t)n_img * 3)); RC(k1.alloc((size_t)n_cam * PXR_KPAD)); RC(X1.alloc((size_t)n_pts * 3));
  RC(scal.alloc(16 + 16 * PXR_LIMBS + 1 + 4));   // ... + one slot whose first 4 bytes are the factorisation's info word (read back with the scalars)   // [0..7] summed over ranks, [8..15] replicated; the limbs of those 16 scalars; the newest linearisation's diagonal statistics
  double* scal_sum = scal.p; double* scal_rep = scal.p + 8;
  const int ldS = n_c + 1;
  double* rhs = S.p + n_c;          // column n_c of S, stride ldS
  DevBuf<double> xsol, linv;
  RC(xsol.alloc(nc1)); RC(linv.alloc(iterative ? 1 : chol_workspace_doubles(n_c)));
  DevBuf<int2> d_col_group; DevBuf<int> d_group_size, d_group_cols;
  DevBuf<double> pcg_u, pcg_mloc, pcg_g, pcg_vec, pcg_scal;
  PcgArgs pcg;
  if (iterative) {
    RC(d_col_group.upload(col_group, st)); RC(d_group_size.upload(group_size, st)); RC(d_group_cols.upload(group_cols, st));
    RC(pcg_u.alloc((size_t)n_pts * 3)); RC(pcg_mloc.alloc((size_t)n_img * DC * DC));
    RC(pcg_g.alloc(group_size.size() * PCG_GS * PCG_GS)); RC(pcg_vec.alloc(5 * nc1));
    RC(pcg_scal.alloc(16 + 8 * ((nc1 + 255) / 256)));   // 8 scalars + the loop's control block + the dot-product partials
  }
  // deterministic mode of the iterative solver: where an image's chunks start (both chunkings), the (image, local column) entries of
  // every reduced-system column in ascending image order, and the buffers of the chunks' partial sums
  DevBuf<int> d_kchunk_ptr, d_schunk_ptr, d_col_ent_ptr;
  DevBuf<int2> d_col_ent;
  DevBuf<double> det_kpart, det_gimg, det_wpart, det_mpart;
  const int det_ne_max = DC * (DC + 1) / 2 + DC;
  if (ctx->deterministic && iterative) {
    auto first_chunks = [&](const std::vector<ImgChunk>& cs) {
      std::vector<int> ptr(n_img + 1, 0);
      for (const ImgChunk& c : cs) ++ptr[c.img + 1];
      for (int i = 0; i < n_img; ++i) ptr[i + 1] += ptr[i];
      return ptr;
    };
    const std::vector<int> kptr = first_chunks(chunks), sptr = first_chunks(schur_chunks);       // (alive until the synchronisation below)
    RC(d_kchunk_ptr.upload(kptr, st)); RC(d_schunk_ptr.upload(sptr, st));
    std::vector<int> ent_ptr(n_c + 1, 0);
    auto col_of = [&](int i, int a) { return a < pose_dim[i] ? pose_off[i] + a : intr_off[image_camera[i]] + (a - pose_dim[i]); };
    for (int i = 0; i < n_img; ++i)
      for (int a = 0; a < pose_dim[i] + intr_dim[image_camera[i]]; ++a) ++ent_ptr[col_of(i, a) + 1];
    for (int c = 0; c < n_c; ++c) ent_ptr[c + 1] += ent_ptr[c];
    std::vector<int2> ent(ent_ptr[n_c]);
    std::vector<int> fill(ent_ptr.begin(), ent_ptr.end() - 1);
    for (int i = 0; i < n_img; ++i)                                          // images ascending: the order of every column's sum
      for (int a = 0; a < pose_dim[i] + intr_dim[image_camera[i]]; ++a) ent[fill[col_of(i, a)]++] = make_int2(i, a);
    RC(d_col_ent_ptr.upload(ent_ptr, st)); RC(d_col_ent.upload(ent, st));
    RC(det_kpart.alloc(std::max<size_t>(1, chunks.size()) * det_ne_max)); RC(det_gimg.alloc((size_t)n_img * DC));
    RC(det_wpart.alloc(std::max<size_t>(1, schur_chunks.size()) * DC)); RC(det_mpart.alloc(std::max<size_t>(1, schur_chunks.size()) * DC * DC));
    PXR_HIP(hipMemsetAsync(det_kpart.p, 0, sizeof(double) * det_kpart.n, st));
    PXR_HIP(hipMemsetAsync(det_wpart.p, 0, sizeof(double) * det_wpart.n, st));
    PXR_HIP(hipMemsetAsync(det_mpart.p, 0, sizeof(double) * det_mpart.n, st));
    PXR_HIP(hipStreamSynchronize(st));        // (the host vectors above leave scope)
  }
  double* diagU = gcd.p; double* gc = gcd.p + nc1;
  DevBuf<double> cb_flags;          // iteration callbacks on several ranks: the ranks' answers, summed
  RC(cb_flags.alloc(2));
  int* d_info = reinterpret_cast<int*>(scal.p + 16 + 16 * PXR_LIMBS);   // zeroed with the scalars (zero_scalars), read back with them

  setup_mark("work buffers allocated");
  double* cur_q = const_cast<double*>(view->d_qvec); double* cur_t = const_cast<double*>(view->d_tvec);
  double* cur_k = const_cast<double*>(view->d_cam_params); double* cur_X = const_cast<double*>(view->d_xyz);
  hipLaunchKernelGGL(k_normalize_q, dim3(nblk(n_img)), dim3(256), 0, st, n_img, cur_q);

  SolveDev dv;
  dv.v = *view;
  dv.pose_off = d_pose_off.p; dv.pose_dim = d_pose_dim.p; dv.tmask = d_tmask.p;
  dv.intr_off = d_intr_off.p; dv.intr_dim = d_intr_dim.p; dv.cmask = d_cmask.p; dv.pt_var = d_pt_var.p;
  dv.scale_c = scale_c.p; dv.scale_p = scale_p.p; dv.n_c = n_c; dv.DC = DC; dv.LS = LS; dv.ldS = ldS;
  pxr_ba_view cand_view = *view;
  cand_view.d_qvec = q1.p; cand_view.d_tvec = t1.p; cand_view.d_cam_params = k1.p; cand_view.d_xyz = X1.p;

  // the collective: the caller's callback if given, else the context's RCCL communicator (pxr_comm_init)
  const bool multi = allreduce != nullptr || (ctx->comm != nullptr && (ctx->nranks > 1 || ctx->force_collective));   // (forced: a one-rank communicator runs the whole multi-rank branch through RCCL -- tests / bench, pxr_comm_force)
  auto ar = [&](double* buf, int64_t count) -> int {
    if (allreduce) {
      if (allreduce(ar_user, buf, count) != 0) return set_error(PXR_EHIP, "pxr_ba_solve: all-reduce callback failed");
      return PXR_OK;
    }
    return comm_allreduce_sum(ctx, buf, count);
  };
  // Replicated quantities (camera step, replicated scalars) are computed by every rank from identical inputs, but
  // with floating-point atomics whose order differs from rank to rank.  To keep the ranks bit-identical -- same
  // parameters, same accept / reject / terminate decisions, hence the same sequence of collectives -- rank 0's copy
  // is broadcast: the others zero theirs and join an all-reduce(sum).  Needs the rank (pxr_comm_init / _set_rank).
  // (Only without the deterministic mode: with it every rank computes the same bits, `bcast` below.)
  DevBuf<double> S_packed;            // the collective's buffer of the direct solver (upper triangle + rhs)
  const int64_t packed_doubles = (int64_t)n_c * (n_c + 3) / 2;
  if (multi && !iterative && n_c > 0) RC(S_packed.alloc((size_t)packed_doubles));
  sum->collective_kib = (multi && !iterative) ? (int)((packed_doubles * 8 + 1023) / 1024) : 0;
  // ---- deterministic mode (the default; pxr_set_deterministic(ctx, 0) / PXR_DETERMINISTIC=0 opt out) -------------------------
  // Everything that is summed over observations, points or ranks is summed as INTEGERS, so that the result does not depend on
  // the order of the adders, on the launch shapes or on how the points are dealt to ranks: an N-rank solve is bit-identical
  // to the one-rank solve.
  //  * matrix / vector slots (U, g_c, S | rhs): fixed point on ONE grid per linearisation, lin_scale = 2^k.  Integer atomics wrap
  //    modulo 2^64, so only the FINAL content of a slot has to fit: |U_ij|, |S_ij| <= max diag(U) (U and the damped S are positive
  //    semi-definite, S <= U) and |g_c|, |rhs| <= sqrt(max diag(U) 2 cost) (Cauchy-Schwarz; rho concave: rho' s <= rho).  The grid
  //    of a linearisation comes from the measured max diag(U) of the previous one with a factor 8 of slack (the first,
  //    Jacobi-scaled one: diag < 1 by construction), and every linearisation is CHECKED (lin_guard below): the finished
  //    diagonal against the bound, and against the trace the chunks measured in floating point -- a diagonal slot that wrapped
  //    is off by a multiple of 2^64 / scale.  A failed check repeats the linearisation on a grid from the measured trace.
  //  * scalars (cost, model cost change, norms): four 40-bit limbs + a bad-addend count per scalar (pxr_device.h), one addend per
  //    observation / point / column.
  //  * several ranks: the integers are all-reduced as integers (ncclInt64; through a sum-of-doubles callback as exact 32-bit
  //    halves), replicated quantities are computed identically by every rank and need no broadcast.
  // (The grids need the Jacobi scaling of the options -- Ceres' and pixsfm's default: without it the large columns of the
  //  unscaled Jacobian, rotations ~1e6, set a grid that drowns the small ones, focal length ~1e-2.  A solve that turns the
  //  scaling off uses floating-point atomics; the iterative solver has its own form of the mode, det_iter below.)
  const bool det_fixed = ctx->deterministic && !iterative && opt->jacobi_scaling != 0;     // fixed-point matrices (direct solver)
  // The iterative solver (> 1000 images) in deterministic mode: ORDERED PARTIAL SUMS instead of fixed point -- every chunk of an
  // image leaves its part in a buffer and the parts are added per image / per column in a fixed order (k_img's kpart, pxr_ba_pcg.hip);
  // the conjugate-gradient vectors span too many orders of magnitude for one grid.  Same bits on every run for a given number of
  // ranks (the ranks' parts are all-reduced as doubles: the result depends on how the points are dealt to ranks, not on the run).
  const bool det_iter = ctx->deterministic && iterative;
  const bool det = det_fixed || det_iter;             // scalars as integer limbs, no rank-0 broadcasts
  sum->accumulation = det_fixed ? 1 : (det_iter ? 2 : 0); sum->initial_us = 0;
  const bool bcast = multi && ctx->nranks > 1 && !det;
  auto from_rank0 = [&](double* buf, int64_t count) -> int {
    if (!bcast) return PXR_OK;
    if (ctx->rank != 0) hipLaunchKernelGGL(k_zero, dim3(nblk(count)), dim3(256), 0, st, count, buf);
    return ar(buf, count);
  };
  struct { long long* p; } slimb;     // 16 scalar slots x PXR_LIMBS integers behind `scal` (one memset clears both): [0..7] summed over the ranks, [8..15] replicated
  slimb.p = reinterpret_cast<long long*>(scal.p + 16);
  long long* const limb_arg = det ? slimb.p : nullptr;     // what the step kernels get: NULL = floating-point atomics into `scal`
  DevBuf<double> i64_halves;          // the callback path's exact 32-bit halves
  const bool native_i64 = allreduce == nullptr;
  if (det && multi && !native_i64) RC(i64_halves.alloc((size_t)2 * std::max<int64_t>({packed_doubles, 2 * (int64_t)nc1 + 8, 16 * PXR_LIMBS})));
  auto ar_i64 = [&](long long* buf, int64_t count) -> int {    // in-place integer all-reduce(sum)
    if (!multi || count <= 0) return PXR_OK;
    if (native_i64) return comm_allreduce_sum_i64(ctx, buf, count);
    hipLaunchKernelGGL(k_i64_split, dim3(nblk(count)), dim3(256), 0, st, count, (const long long*)buf, i64_halves.p);
    RC(ar(i64_halves.p, 2 * count));
    hipLaunchKernelGGL(k_i64_join, dim3(nblk(count)), dim3(256), 0, st, count, (const double*)i64_halves.p, buf);
    return PXR_OK;
  };
  auto zero_scalars = [&]() -> int {
    PXR_HIP(hipMemsetAsync(scal.p, 0, sizeof(double) * (16 + 16 * PXR_LIMBS + 1), st));   // scalars, limbs, the info word (the statistics behind them stay)
    return PXR_OK;
  };
  struct { double* p; } lin_stats;    // {max, sum, min of diag(U), trace} of the newest linearisation (k_finish_and_stats), behind the 16 scalars
  constexpr int kScalAll = 16 + 16 * PXR_LIMBS + 1 + 4;
  lin_stats.p = scal.p + (kScalAll - 4);
  // read back into PINNED host memory: an asynchronous copy into pageable memory (a stack array) is staged by the runtime and
  // blocks the host until it is done -- two such copies per attempt (scalars, factorisation info) left the GPU idle for ~25 us
  // each (profiles/r6_lm_timeline.txt)
  if (!ctx->h_readback) PXR_HIP(hipHostMalloc(&ctx->h_readback, 4096, hipHostMallocDefault));
  static_assert(sizeof(double) * kScalAll <= 4096, "read-back buffer");
  double* const h_scal = static_cast<double*>(ctx->h_readback);
  std::memset(h_scal, 0, sizeof(double) * kScalAll);
  double* const h_lin_stats = h_scal + (kScalAll - 4);
  const int* const h_info_word = reinterpret_cast<const int*>(h_scal + 16 + 16 * PXR_LIMBS);
  const bool spin_wait = std::getenv("PXR_BLOCKING_WAIT") == nullptr;
  auto read_scal = [&](double* h16) -> int {
    if (det) {
      RC(ar_i64(slimb.p, 8 * PXR_LIMBS));
      hipLaunchKernelGGL(k_limbs_finish, dim3(1), dim3(64), 0, st, (const long long*)slimb.p, 0xffffu, scal.p);
    } else {
      hipLaunchKernelGGL(k_limbs_finish, dim3(1), dim3(64), 0, st, (const long long*)slimb.p, (1u << 0) | (1u << 4), scal.p);   // the costs
      RC(ar(scal_sum, 8));
      RC(from_rank0(scal_rep, 8));
    }
    PXR_HIP(hipMemcpyAsync(h_scal, scal.p, sizeof(double) * kScalAll, hipMemcpyDeviceToHost, st));     // the scalars ... the statistics: one copy of 800 bytes
    // the one host synchronisation of an LM attempt: polled (the blocking wait of hipStreamSynchronize sleeps on an interrupt and
    // wakes tens of microseconds late -- once per attempt, with the GPU idle meanwhile); PXR_BLOCKING_WAIT=1 restores it
    if (spin_wait) {
      PXR_HIP(hipEventRecord(ctx->ev_sync, st));
      hipError_t q;
      while ((q = hipEventQuery(ctx->ev_sync)) == hipErrorNotReady) {}
      PXR_HIP(q);
    } else PXR_HIP(hipStreamSynchronize(st));
    std::memcpy(h16, h_scal, sizeof(double) * 16);
    return PXR_OK;
  };
  DevBuf<double> det_part, chunk_trace;
  RC(det_part.alloc((size_t)n_pts + 8));          // the inner iterations' per-point costs (every mode)
  RC(chunk_trace.alloc(chunks.size() + 1));
  // the grid: entries bounded by 8 max(md, sqrt(2 md cost)) fit 62 bits (md: the bound on diag(U) the grid is made for).  (A grid
  // of 2^51 units with a one-addition rounding was tried: it saves nothing measurable -- the Schur kernel waits on its gathers --
  // and the 11 lost bits broke the parity of ill-conditioned reduced systems with the oracle.  What the mode costs: `lm.nondeterministic` in
  // profiles/r5_bench_n1.json, 2.37 against 2.45 ms per iteration.)
  auto det_scale_for = [](double md, double cost_now) {
    const double bound = 8.0 * std::max({md, std::sqrt(2.0 * md * std::max(cost_now, 0.0)), 1e-300});
    return std::ldexp(1.0, 62 - (int)std::ceil(std::log2(bound)));
  };
  double lin_md = 1.0;                // the diag(U) bound the CURRENT grid was made for
  // pxr_set_gram_cache: the records from cached Gram matrices of the stencils instead of from the texels (pxr_ba_gram.hip)
  // (default on since round 5: pinned against the reference functor's vectors at 1e-5, tests/test_gram_cache_gpu.py.  check_bounds
  //  needs no care: with reference descriptors the functor ignores the bounds check, feature_reference.h:128-136.)
  // (InterpolationConfig.use_float_simd asks for the reference's ALL-fp32 splines: the Gram-matrix paths -- exact fp64 algebra -- would
  // silently compute something finer; a solve with that flag keeps the texel kernels, which have the fp32 arithmetic bit for bit)
  bool gram_cache = ctx->gram_cache && gram_eval_supported(arena, view) && cfg->use_float_simd == 0;
  // The Gram-matrix kernel of the inner iterations keeps its matrices in the same cache from call to call (it writes back what
  // it builds), whether or not the LM loop evaluates from them: the same numbers as without a cache, fewer builds.  (That kernel
  // is built without the six extended camera models -- their forward-mode duals cost ~100 registers: a problem that uses one
  // keeps the packed kernel for every point.)
  bool gram_inner = opt->use_inner_iterations != 0 && arena->dtype != PXR_F64 && (arena->C == 128 || arena->C == 64) && !getenv("PXR_INNER_OLD") &&
                    !getenv("PXR_INNER_PACKED") && cfg->use_float_simd == 0;
  for (int c = 0; c < n_cam; ++c) gram_inner = gram_inner && cam_model[c] <= PXR_OPENCV;
  bool inner_cache = gram_inner && gram_eval_supported(arena, view) && !getenv("PXR_INNER_NO_CACHE");
  GramCache gram;
  if ((gram_cache || inner_cache) && gram_eval_prepare(ctx, arena, view, &gram) != PXR_OK) {
    // no memory for the cache (1.4 KB per observation, next to the solver's own buffers): not an error of this solve -- a problem
    // that fitted before the cache became the default must still run.  The LM loop evaluates from the texels (the exact-order
    // kernel), the inner iterations build their matrices at every call.
    gram_cache = false; inner_cache = false;
    (void)set_error(PXR_OK, "");    // pxr_last_error() must not keep the allocation's message
  }
  setup_mark("Gram-matrix cache prepared");
  bool gram_warm = false;               // the cache holds every observation's matrices (after the first evaluation / inner call)
  int n_evaluations = 0;
  // The initial point is evaluated by the exact-order kernel (below), candidates from the cache: until a step is accepted the
  // current and the candidate cost come from two evaluations that differ by the rounding of the reference's fp32 pass (~1e-9
  // relative).  A solve that starts AT an optimum would decide its first steps on that difference: after the first rejected
  // step the current point is evaluated again, from the cache (its cells are mostly the candidate's: cheap), so that cost
  // changes compare like with like.
  bool cur_is_exact = true;
  auto evaluate = [&](const pxr_ba_view& v, double* rec) -> int {   // rec + cost into scal_sum[0]
    // (the evaluation at the INITIAL point takes the exact-order kernel also with the cache on: the first trust-region step is
    // usually the largest of the solve -- at configs[2] 95 % of the projections leave their cell -- so matrices built there
    // would be built again at once, and 1.5 ms of builds cost more than 0.8 ms of texels)
    if (gram_cache && n_evaluations > 0) { RC(gram_evaluate(ctx, arena, &v, cfg, gram, rec)); gram_warm = true; }
    else RC(ba_eval_with_cost(ctx, arena, &v, cfg, 1, rec, nullptr, nullptr, nullptr, nullptr, nullptr));
    ++n_evaluations;
    hipLaunchKernelGGL(k_cost_limbs, dim3(1024), dim3(256), 0, st, (const double*)rec, n_obs, *loss, slimb.p + 0 * PXR_LIMBS);
    LAUNCH_CHECK("cost");
    return PXR_OK;
  };
  // linearise at the CURRENT parameters from record buffer `rec` (lin_scale: the fixed-point grid of U and g_c in
  // deterministic mode, 0 otherwise)
  double lin_scale = 0.0;
  auto linearize = [&](const double* rec) -> int {
    hipLaunchKernelGGL(k_jac, dim3((unsigned)((n_obs + JAC_THREADS - 1) / JAC_THREADS)), dim3(JAC_THREADS),
                       sizeof(double) * (JAC_THREADS / 64) * 64 * (LS + 3 * DC + 1), st, dv, rec, *loss, L.p, W.p);
    hipLaunchKernelGGL(k_point, dim3(nblk(n_pts)), dim3(256), 0, st, dv, d_pt_ptr.p, d_pt_obs.p, L.p, V.p, gp.p);
    PXR_HIP(hipMemsetAsync(U.p, 0, sizeof(double) * U.n, st));
    PXR_HIP(hipMemsetAsync(gcd.p, 0, sizeof(double) * gcd.n, st));
    if (n_c > 0 && !chunks.empty()) {
      hipLaunchKernelGGL(k_img, dim3((unsigned)chunks.size()), dim3(256), sizeof(double) * ((size_t)IMG_BATCH * LS + 256), st, dv,
                         d_chunks.p, d_img_obs.p, L.p, U.p, gc, iterative ? 1 : 0, lin_scale, lin_scale != 0.0 ? chunk_trace.p : (double*)nullptr,
                         det_iter ? det_kpart.p : (double*)nullptr, det_ne_max);
      if (det_iter) {      // U blocks, diag(U) and g_c as ordered sums of the chunks' parts
        RC(pcg_blocks_from_partials(st, dv, d_kchunk_ptr.p, det_kpart.p, det_ne_max, U.p, det_gimg.p, d_col_ent_ptr.p, d_col_ent.p, diagU, gc));
        LAUNCH_CHECK("linearize kernels");
        RC(ar(gcd.p, 2 * (int64_t)nc1));
        return PXR_OK;
      }
      if (lin_scale != 0.0) {
        // the integers of diag(U) and g_c (+ the trace) are summed over the ranks, THEN everything becomes doubles again
        hipLaunchKernelGGL(k_diag_and_trace, dim3(1), dim3(1024), 0, st, n_c, (const double*)U.p, diagU, (const double*)chunk_trace.p, (int)chunks.size(),
                           reinterpret_cast<long long*>(gcd.p + 2 * nc1));
        RC(ar_i64(reinterpret_cast<long long*>(gcd.p), 2 * (int64_t)nc1 + PXR_LIMBS));
        // (U itself stays in fixed point: only the Schur complement reads it, as integers)
        hipLaunchKernelGGL(k_finish_and_stats, dim3(1), dim3(1024), 0, st, n_c, 2 * (int)nc1, gcd.p, lin_scale,
                           reinterpret_cast<const long long*>(gcd.p + 2 * nc1), lin_stats.p);
        LAUNCH_CHECK("linearize kernels");
        return PXR_OK;
      }
      if (iterative) RC(pcg_diag_from_blocks(st, dv, U.p, diagU));
      else hipLaunchKernelGGL(k_extract_diag, dim3(nblk(n_c)), dim3(256), 0, st, n_c, U.p, diagU);
    }
    LAUNCH_CHECK("linearize kernels");
    RC(ar(gcd.p, 2 * (int64_t)nc1));   // global diag(U) and g_c
    return PXR_OK;
  };
  // the overflow guard: did every slot of the newest linearisation fit its grid?  (stats: lin_stats, read with the scalars or here)
  auto lin_fits = [&](const double* stt, double md_made_for) {
    if (n_c == 0 || chunks.empty()) return true;
    const double mx = stt[0], sm = stt[1], mn = stt[2], tr = stt[3];
    if (!(std::isfinite(mx) && std::isfinite(sm) && std::isfinite(tr))) return false;
    if (mn < 0.0) return false;                                               // a diagonal slot wrapped into the sign bit
    if (std::fabs(tr - sm) > std::ldexp(1.0, 61) / lin_scale) return false;   // ... or all the way round
    return mx <= 8.0 * md_made_for;                                           // the off-diagonal, Schur and gradient bounds hold
  };
  // linearise on a grid made for diag(U) <= 8 md_guess; repeat on a grid from the measured trace until the check passes
  // (synchronises: used for the first two linearisations of a solve and on the -- never yet observed -- failure path)
  bool lin_not_finite = false;        // set by linearize_checked: the Jacobian at the current point is not finite
  auto linearize_checked = [&](const double* rec, double md_guess, double cost_now, bool refine = false) -> int {
    if (!det_fixed) return linearize(rec);
    for (int attempt = 0; attempt < 8; ++attempt) {
      lin_md = md_guess;
      lin_scale = det_scale_for(lin_md, cost_now);
      RC(linearize(rec));
      PXR_HIP(hipMemcpyAsync(h_lin_stats, lin_stats.p, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
      PXR_HIP(hipStreamSynchronize(st));
      if (lin_fits(h_lin_stats, lin_md)) {
        // `refine`: this linearisation is SOLVED with (no Jacobi scaling follows) -- a guess far above the measured diagonal
        // wasted resolution, once more on a grid made for the measurement
        if (refine && attempt == 0 && h_lin_stats[0] * 64.0 < md_guess) { md_guess = std::max(h_lin_stats[0], 1e-300); continue; }
        return PXR_OK;
      }
      // the trace was measured in floating point and bounds every diagonal entry.  NaN / Inf: the Jacobian itself is not finite --
      // [upstream] "Residual and Jacobian evaluation failed": the solve TERMINATES with FAILURE at the last accepted point, it is
      // not an error of the call (this used to return PXR_EINVAL)
      if (!(std::isfinite(h_lin_stats[3]) && h_lin_stats[3] >= 0.0)) { lin_not_finite = true; return PXR_OK; }
      md_guess = std::max(h_lin_stats[3], 2.0 * md_guess);
    }
    return set_error(PXR_EINVAL, "pxr_ba_solve: the fixed-point grid of the deterministic mode could not be fitted");
  };
  auto refresh_damping = [&]() -> int {  // LevenbergMarquardtStrategy: diagonal clamped to [min, max]
    if (n_c > 0) hipLaunchKernelGGL(k_clamp, dim3(nblk(n_c)), dim3(256), 0, st, (int64_t)n_c, diagU, opt->min_lm_diagonal, opt->max_lm_diagonal, damp_c.p);
    hipLaunchKernelGGL(k_point_diag, dim3(nblk(n_pts)), dim3(256), 0, st, n_pts, V.p, Vd0.p);
    hipLaunchKernelGGL(k_clamp, dim3(nblk(n_pts * 3)), dim3(256), 0, st, n_pts * 3, Vd0.p, opt->min_lm_diagonal, opt->max_lm_diagonal, Vd0.p);
    LAUNCH_CHECK("damping kernels");
    return PXR_OK;
  };

  if (iterative) {
    pcg.st = st; pcg.dv = dv;
    pcg.chunks = d_schur_chunks.p; pcg.n_chunks = (int)schur_chunks.size(); pcg.so = d_so.p;
    pcg.pt_ptr = d_pt_ptr.p; pcg.part_obs = d_part_obs.p; pcg.obs_cols = d_obs_cols.p;
    pcg.W = W.p; pcg.T = T.p; pcg.gp = gp.p; pcg.gc = gc; pcg.damp_c = damp_c.p; pcg.Ublk = U.p;
    pcg.n_groups = (int)group_size.size(); pcg.col_group = d_col_group.p; pcg.group_size = d_group_size.p;
    pcg.group_cols = d_group_cols.p;
    pcg.u = pcg_u.p; pcg.Mloc = pcg_mloc.p; pcg.Gm = pcg_g.p;
    pcg.x = xsol.p; pcg.r = pcg_vec.p; pcg.p = pcg_vec.p + nc1; pcg.q = pcg_vec.p + 2 * nc1; pcg.z = pcg_vec.p + 3 * nc1;
    pcg.b = pcg_vec.p + 4 * nc1; pcg.cgs = pcg_scal.p; pcg.cg_part = pcg_scal.p + 16; pcg.d_fail = d_info;
    pcg.det = det_iter;
    if (det_iter) {
      pcg.chunk_ptr = d_schunk_ptr.p; pcg.col_ent_ptr = d_col_ent_ptr.p; pcg.col_ent = d_col_ent.p;
      pcg.wpart = det_wpart.p; pcg.mpart = det_mpart.p;
    }
  }
  const std::function<int(double*, int64_t)> ar_fn = ar;
  // gradient_tolerance [upstream]: max-norm of the gradient in the unscaled variables, over ALL ranks
  auto gradient_below_tolerance = [&](bool* below) -> int {
    double h[16];
    RC(zero_scalars());
    if (n_c > 0) hipLaunchKernelGGL(k_count_above, dim3(nblk(n_c)), dim3(256), 0, st, (int64_t)n_c, gc, scale_c.p, opt->gradient_tolerance, scal_rep + 4,
                                    det ? slimb.p + (8 + 4) * PXR_LIMBS : (long long*)nullptr);
    hipLaunchKernelGGL(k_count_above, dim3(nblk(n_pts * 3)), dim3(256), 0, st, n_pts * 3, gp.p, scale_p.p, opt->gradient_tolerance, scal_sum + 5,
                       det ? slimb.p + 5 * PXR_LIMBS : (long long*)nullptr);
    RC(read_scal(h));
    *below = h[5] == 0.0 && h[12] == 0.0;
    return PXR_OK;
  };

  // the inner iterations' tables (a pass over the points on the host, two small uploads): set-up, like the observation lists
  struct InnerListsOwner { InnerLists l; ~InnerListsOwner() { free_inner_lists(&l); } } inner_lists;
  if (gram_inner) {
    const int64_t* h_pt_ptr = pt_cnt.data();                // (general path: pt_cnt holds the prefix sums by now)
    std::vector<int64_t> pageable;
    void* wave_stage = nullptr;                              // pinned room for the table the host packs (uploaded from where it is built)
    const size_t ptr_bytes = (sizeof(int64_t) * ((size_t)n_pts + 1) + 255) & ~(size_t)255, wave_bytes = inner_wave_stage_bytes(n_pts);
    if (device_lists) {                                      // fast path: the offsets come back once, through pinned memory
      int64_t* stage = (ptr_bytes + wave_bytes) <= ((size_t)1 << 28) ? static_cast<int64_t*>(setup_staging(ctx, ptr_bytes + wave_bytes)) : nullptr;
      if (stage) wave_stage = reinterpret_cast<char*>(stage) + ptr_bytes;
      else { pageable.resize((size_t)n_pts + 1); stage = pageable.data(); }
      PXR_HIP(hipMemcpyAsync(stage, d_pt_ptr.p, sizeof(int64_t) * ((size_t)n_pts + 1), hipMemcpyDeviceToHost, st));
      PXR_HIP(hipStreamSynchronize(st));
      h_pt_ptr = stage;
    }
    RC(make_inner_lists(ctx, h_pt_ptr, n_pts, wave_stage, wave_stage ? wave_bytes : 0, view, d_pt_ptr.p, d_pt_obs.p, &inner_lists.l));
  }
  setup_mark("inner-iteration tables");
  const auto t_loop0 = std::chrono::steady_clock::now();
  sum->setup_ms = std::chrono::duration<double, std::milli>(t_loop0 - t_setup0).count();
  sum->num_camera_unknowns = n_c; sum->num_point_unknowns = 3 * n_pvar;
  sum->iterations = 0; sum->num_successful = 0; sum->termination = PXR_TERM_NO_CONVERGENCE;
  // LDS-privatised Schur contraction: column tile so that DC x CT doubles fit in 128 KiB of LDS
  const bool use_lds_schur = std::getenv("PXR_SCHUR_GLOBAL_ATOMICS") == nullptr;
  int CT = n_c > 0 ? std::min(n_c, (int)((128 * 1024 / 8 - DC) / DC)) : 1;
  int n_ctiles = n_c > 0 ? (n_c + CT - 1) / CT : 1;
  if (const char* e = std::getenv("PXR_SCHUR_CTILES")) n_ctiles = std::max(n_ctiles, std::min(std::max(1, n_c), std::atoi(e)));   // A/B knob (profiles/r6_schur_tiles.txt)
  CT = n_c > 0 ? (n_c + n_ctiles - 1) / n_ctiles : 1;       // balance the tiles
  const size_t schur_shmem = sizeof(double) * ((size_t)DC * CT + DC + 3 * 1024);   // tile, right-hand side, the lane groups' Y rows
  if (use_lds_schur && n_c > 0)
    PXR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(DC <= 8 ? k_schur_lds<8> : (DC <= 16 ? k_schur_lds<16> : k_schur_lds<32>)),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)schur_shmem));
  PXR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_jac), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(sizeof(double) * (JAC_THREADS / 64) * 64 * (LS + 3 * DC + 1))));
  const bool verbose = std::getenv("PXR_VERBOSE") != nullptr;
  const bool phase_timing = std::getenv("PXR_PHASE_TIMING") != nullptr;   // adds a stream sync per phase
  double ph_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto ph_t0 = std::chrono::steady_clock::now();
  auto phase = [&](int k) {
    if (!phase_timing) return;
    (void)hipStreamSynchronize(st);
    const auto now = std::chrono::steady_clock::now();
    ph_ms[k] += std::chrono::duration<double, std::milli>(now - ph_t0).count();
    ph_t0 = now;
  };

  // ---- iteration 0: evaluate, Jacobi scaling, linearise ------------------------------------------------
  double hs[16];
  double* rec_cur = rec_a.p; double* rec_cand = rec_b.p;
  RC(zero_scalars());
  RC(evaluate(dv.v, rec_cur));
  RC(read_scal(hs));
  double cost = hs[0];
  sum->initial_cost = cost;
  if (!std::isfinite(cost)) {   // [upstream] "Initial residual and Jacobian evaluation failed" (e.g. check_bounds)
    sum->final_cost = cost; sum->termination = PXR_TERM_FAILURE;
    sum->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count();
    return PXR_OK;
  }
  hipLaunchKernelGGL(k_fill, dim3(nblk(nc1)), dim3(256), 0, st, (int64_t)nc1, 1.0, scale_c.p);
  hipLaunchKernelGGL(k_fill, dim3(nblk(n_pts * 3)), dim3(256), 0, st, n_pts * 3, 1.0, scale_p.p);
  // (deterministic mode: the UNSCALED pass only yields diag(U) for the Jacobi scaling and has no a-priori bound: a first guess
  //  of 2^32 for its diagonal -- unit-norm descriptors give 1e3 .. 1e6 -- and the checked retry from the measured trace otherwise)
  RC(linearize_checked(rec_cur, std::ldexp(1.0, 32), cost, !opt->jacobi_scaling));
  auto fail_not_finite = [&]() -> int {
    sum->final_cost = cost; sum->final_radius = opt->initial_radius; sum->termination = PXR_TERM_FAILURE;
    sum->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count();
    return PXR_OK;
  };
  if (lin_not_finite) return fail_not_finite();
  if (opt->jacobi_scaling) {   // 1 / (1 + sqrt(diag(J~^T J~))), fixed for the whole solve
    if (n_c > 0) hipLaunchKernelGGL(k_jacobi_scale, dim3(nblk(n_c)), dim3(256), 0, st, (int64_t)n_c, diagU, scale_c.p);
    hipLaunchKernelGGL(k_point_diag, dim3(nblk(n_pts)), dim3(256), 0, st, n_pts, V.p, Vd0.p);
    hipLaunchKernelGGL(k_jacobi_scale, dim3(nblk(n_pts * 3)), dim3(256), 0, st, n_pts * 3, Vd0.p, scale_p.p);
    RC(linearize_checked(rec_cur, 1.0 / 8.0, cost));      // scaled columns have norm < 1: diag(U) < 1 = 8 x 1/8
    if (lin_not_finite) return fail_not_finite();
  }
  if (opt->gradient_tolerance > 0.0) {   // [upstream] the test is also made at iteration 0
    bool below = false;
    RC(gradient_below_tolerance(&below));
    if (below) {
      sum->termination = PXR_TERM_CONVERGENCE; sum->final_cost = cost; sum->final_radius = opt->initial_radius;
      sum->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count();
      return PXR_OK;
    }
  }
  double radius = opt->initial_radius, decrease_factor = 2.0;
  int invalid = 0;
  bool reuse_diag = false;
  bool inner_enabled = opt->use_inner_iterations != 0, inner_useful = false;
  PXR_HIP(hipStreamSynchronize(st));         // (already idle in the deterministic mode: linearize_checked synchronises)
  sum->initial_us = (int32_t)std::min<double>(2e9, 1e3 * std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count());
  // ceres::IterationCallback (pxr_set_iteration_callback): 1 = SOLVER_ABORT, 2 = SOLVER_TERMINATE_SUCCESSFULLY
  // Several ranks: a callback installed on SOME ranks only (rank 0 logging, say) must not put the ranks' collectives out of
  // step -- whether any rank has one is agreed on once here; if so, every rank joins the per-iteration exchange of answers
  // (a rank without a callback answers "continue").
  bool any_cb = ctx->iter_cb != nullptr;
  if (multi) {
    double h = any_cb ? 1.0 : 0.0;
    PXR_HIP(hipMemcpyAsync(cb_flags.p, &h, sizeof(h), hipMemcpyHostToDevice, st));
    RC(ar(cb_flags.p, 1));
    PXR_HIP(hipMemcpyAsync(&h, cb_flags.p, sizeof(h), hipMemcpyDeviceToHost, st));
    PXR_HIP(hipStreamSynchronize(st));
    any_cb = h > 0.0;
  }
  auto notify = [&](int iteration, bool valid, bool successful, double cost_now, double change, double rel, double radius_next, double step) -> int {
    if (!any_cb) return 0;
    int rc = 0;
    if (ctx->iter_cb) {
      pxr_iteration_summary is;
      is.iteration = iteration; is.step_is_valid = valid; is.step_is_successful = successful; is.cost = cost_now; is.cost_change = change;
      is.relative_decrease = rel; is.trust_region_radius = radius_next; is.step_norm = step;
      rc = ctx->iter_cb(&is, ctx->iter_user);
    }
    if (multi) {
      // every rank must leave the loop at the same iteration (the next collective would hang otherwise): the answers are
      // summed over the ranks, an abort anywhere aborts everywhere, else a termination request anywhere terminates
      double h[2] = {rc == 1 ? 1.0 : 0.0, rc == 2 ? 1.0 : 0.0};
      if (hipMemcpyAsync(cb_flags.p, h, sizeof(h), hipMemcpyHostToDevice, st) != hipSuccess || ar(cb_flags.p, 2) != PXR_OK ||
          hipMemcpyAsync(h, cb_flags.p, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return 1;
      rc = h[0] > 0.0 ? 1 : (h[1] > 0.0 ? 2 : 0);
    }
    return rc;
  };
  auto user_stop = [&](int rc) {   // true: leave the loop
    if (rc == 1) { sum->termination = PXR_TERM_FAILURE; return true; }
    if (rc == 2) { sum->termination = PXR_TERM_CONVERGENCE; return true; }
    return false;
  };
  if (user_stop(notify(0, false, false, cost, 0.0, 0.0, radius, 0.0))) {
    sum->final_cost = cost; sum->final_radius = radius;
    sum->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count();
    return PXR_OK;
  }

  while (true) {
    if (sum->iterations >= opt->max_iterations) { sum->termination = PXR_TERM_NO_CONVERGENCE; break; }
    if (radius < opt->min_radius) { sum->termination = PXR_TERM_CONVERGENCE; break; }
    ++sum->iterations;
    phase(7);
    if (!reuse_diag) RC(refresh_damping());
    const double inv_radius = 1.0 / radius;
    // point elimination + reduced camera system
    hipLaunchKernelGGL(k_pinv, dim3(nblk(n_pts)), dim3(256), 0, st, n_pts, d_pt_var.p, V.p, Vd0.p, inv_radius, T.p, scal.p, 16 + 16 * PXR_LIMBS + 1);   // (+ zero_scalars' job)
    if (n_pts == 0) RC(zero_scalars());
    bool ok = true;
    double inexact_correction = 0.0;
    if (n_c > 0 && iterative) {
      // ITERATIVE_SCHUR: preconditioned conjugate gradients on the implicit Schur complement (pxr_ba_pcg.hip)
      PcgResult pr;
      RC(pcg_solve(pcg, inv_radius, opt, ar_fn, &pr));
      phase(2);
      sum->linear_iterations += pr.iterations;
      if (!pr.ok) ok = false;                                 // LINEAR_SOLVER_FAILURE: invalid step
      // The camera step only approximately solves S x = b; the points are eliminated exactly for that x.  With
      // e = b - S x:  model cost change = delta.(D^2 delta - g) / 2 + x.e / 2
      inexact_correction = 0.5 * pr.x_dot_r;
      PXR_HIP(hipMemsetAsync(d_info, 0, sizeof(int), st));    // (the CG loop's failure flag lives in the info word: reported through pr.ok)
      hipLaunchKernelGGL(k_finish_camera_step, dim3(nblk(n_c)), dim3(256), 0, st, n_c, xsol.p, gc, damp_c.p, inv_radius, delta_c.p, scal_rep, limb_arg);
      RC(from_rank0(delta_c.p, n_c));
    } else if (n_c > 0) {
      const double schur_scale = det_fixed ? lin_scale : 0.0;    // the SAME grid as U's: re-quantising the finished U is exact
      hipLaunchKernelGGL(k_copy_upper_add_diag, dim3(nblk((int64_t)n_c * ldS)), dim3(256), 0, st, n_c, U.p, (const double*)nullptr, 0.0, (const double*)nullptr, S.p, 1, schur_scale);
      if (use_lds_schur) {
#define SCHUR_LAUNCH(GG)                                                                                              \
  hipLaunchKernelGGL(k_schur_lds<GG>, dim3((unsigned)schur_chunks.size(), (unsigned)n_ctiles), dim3(1024), schur_shmem, st, dv, \
                     d_schur_chunks.p, d_so.p, d_part_obs.p, d_obs_cols.p, W.p, T.p, gp.p, CT, S.p, rhs, schur_scale)
        if (DC <= 8) SCHUR_LAUNCH(8); else if (DC <= 16) SCHUR_LAUNCH(16); else SCHUR_LAUNCH(32);
#undef SCHUR_LAUNCH
      } else {
        hipLaunchKernelGGL(k_schur, dim3(nblk(n_obs * DC)), dim3(256), 0, st, dv, d_pt_ptr.p, d_pt_obs.p, W.p, T.p, gp.p, S.p, rhs, schur_scale);
      }
      LAUNCH_CHECK("schur kernels");
      phase(0);
      if (multi) {       // sum of U_local - Schur_local and of -Y g_p over the ranks: packed upper triangle + rhs
        hipLaunchKernelGGL(k_pack_upper, dim3((unsigned)n_c), dim3(256), 0, st, n_c, S.p, S_packed.p);
        if (schur_scale != 0.0) RC(ar_i64(reinterpret_cast<long long*>(S_packed.p), packed_doubles));   // the fixed-point integers
        else RC(ar(S_packed.p, packed_doubles));
        hipLaunchKernelGGL(k_unpack_upper, dim3((unsigned)n_c), dim3(256), 0, st, n_c, S_packed.p, S.p);
        LAUNCH_CHECK("pack / unpack of the reduced camera system");
      }
      phase(1);
      // the fixed-point slots back to doubles (deterministic mode), rhs += g_c (global), S += D_c / radius: one pass
      hipLaunchKernelGGL(k_copy_upper_add_diag, dim3(nblk((int64_t)n_c * ldS)), dim3(256), 0, st, n_c, U.p, damp_c.p, inv_radius, gc, S.p, 0, schur_scale);
      // row-major upper == column-major lower; the pivot check is read back with the scalars of
      // this attempt (no extra host synchronisation): a failed factorisation = invalid step.
      RC(chol_factor_solve(st, S.p, n_c, d_info, linv.p, xsol.p, /*zero_info=*/false));     // (zero_scalars cleared the info word)
      phase(2);
      hipLaunchKernelGGL(k_finish_camera_step, dim3(nblk(n_c)), dim3(256), 0, st, n_c, xsol.p, gc, damp_c.p, inv_radius, delta_c.p, scal_rep, limb_arg);
      RC(from_rank0(delta_c.p, n_c));
    }
    phase(3);
    double model_cost_change = 0, cand_cost = 0, step_norm = 0, x_norm = 0;
    if (ok) {
#define BACKSUB_LAUNCH(GG)                                                                                                          \
  hipLaunchKernelGGL(k_backsub<GG>, dim3(nblk((n_pts * GG + BACKSUB_PPG - 1) / BACKSUB_PPG)), dim3(256), 0, st, dv, d_pt_ptr.p, d_part_obs.p, d_obs_cols.p, W.p, T.p, gp.p, \
                     delta_c.p, Vd0.p, inv_radius, delta_p.p, scal_sum, limb_arg)
      if (DC <= 8) BACKSUB_LAUNCH(8); else if (DC <= 16) BACKSUB_LAUNCH(16); else BACKSUB_LAUNCH(32);
#undef BACKSUB_LAUNCH
      ParamPtrs out{q1.p, t1.p, k1.p, X1.p};
      const int64_t upd_blocks = nblk(((int64_t)n_img + n_cam + n_pts + UPDATE_IPT - 1) / UPDATE_IPT);
      hipLaunchKernelGGL(k_update, dim3((unsigned)upd_blocks), dim3(256), 0, st, dv, delta_c.p, delta_p.p, out, scal_rep, scal_sum, limb_arg);
      LAUNCH_CHECK("step kernels");
      const bool do_inner = inner_enabled;
      if (do_inner) {   // DoInnerIterationsIfNeeded [upstream]: refine every variable point of the candidate on its own
        // the cost at the unrefined candidate: per-point values, summed as limbs in every mode (one floating-point atomic per
        // point on ONE address -- 200 000 of them at configs[2] -- serialised for 0.5-1 ms of the 2-4 ms call:
        // profiles/r4_inner_cost_atomic.txt)
        PXR_HIP(hipMemsetAsync(det_part.p, 0, sizeof(double) * (size_t)n_pts, st));   // points without observations stay 0
        RC(launch_inner_iterations(ctx, arena, &cand_view, cfg, loss, d_pt_ptr.p, d_pt_obs.p, d_pt_var.p, scal_sum + 4, gram_inner ? &inner_lists.l : nullptr,
                                   det_part.p, (gram_cache || inner_cache) ? &gram : nullptr, gram_warm));
        gram_warm = true;
        hipLaunchKernelGGL(k_limb_accumulate, dim3(256), dim3(256), 0, st, (const double*)det_part.p, n_pts, slimb.p + 4 * PXR_LIMBS);
        PXR_HIP(hipMemsetAsync(scal_sum + 2, 0, sizeof(double), st));   // point part of |x - candidate|^2 after refinement
        PXR_HIP(hipMemsetAsync(slimb.p + 2 * PXR_LIMBS, 0, sizeof(long long) * PXR_LIMBS, st));
        hipLaunchKernelGGL(k_point_step_norm, dim3(nblk(n_pts)), dim3(256), 0, st, n_pts, d_pt_var.p, cur_X, X1.p, scal_sum + 2,
                           det ? slimb.p + 2 * PXR_LIMBS : (long long*)nullptr);
        LAUNCH_CHECK("inner iteration kernels");
      }
      phase(4);
      RC(evaluate(cand_view, rec_cand));
      RC(read_scal(hs));
      const int h_info = *h_info_word;
      if (det_fixed && !lin_fits(h_lin_stats, lin_md)) {
        // the overflow guard: a slot of the linearisation this iteration was computed from did not fit its grid -- repeat the
        // linearisation on a grid from the measured trace and the iteration with it (state untouched: radius, damping, counts)
        RC(linearize_checked(rec_cur, std::max(h_lin_stats[3], 2.0 * lin_md), cost));
        if (lin_not_finite) { sum->termination = PXR_TERM_FAILURE; break; }
        --sum->iterations;
        continue;
      }
      if (n_c > 0 && h_info != 0) ok = false;
      phase(5);
      cand_cost = hs[0];
      model_cost_change = 0.5 * (hs[1] + hs[8]) + inexact_correction;   // 0.5 * delta.(D^2 delta - g) == -(J d).(r + J d / 2)
      step_norm = std::sqrt(hs[2] + hs[9]);
      x_norm = std::sqrt(hs[3] + hs[10]);
      if (!(model_cost_change > 0.0) || !std::isfinite(model_cost_change)) ok = false;
      // A candidate that cannot be evaluated (check_bounds, a point behind a camera) is NOT an invalid step:
      // [upstream] sets candidate_cost = DBL_MAX and the step is rejected through the relative decrease.
      if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
      inner_useful = false;
      if (ok && do_inner && cand_cost < std::numeric_limits<double>::max()) {
        const double cost_before = hs[4];                 // cost at the candidate before the inner iterations
        model_cost_change += cost_before - cand_cost;
        inner_useful = cand_cost < cost;
        inner_enabled = (1.0 - cand_cost / cost_before) > opt->inner_iteration_tolerance;
      }
    }
    if (verbose)
      fprintf(stderr, "[pxr_ba_solve] it %3d cost %.9e cand %.9e mcc %.3e radius %.3e |dx| %.3e %s\n", sum->iterations,
              cost, cand_cost, model_cost_change, radius, step_norm, ok ? "" : "INVALID");
    auto reevaluate_current = [&]() -> int {          // rec_cur and `cost` through the cache path
      if (!(gram_cache && cur_is_exact && n_evaluations > 1)) return PXR_OK;
      double h2[16];
      RC(zero_scalars());
      RC(evaluate(dv.v, rec_cur));
      RC(read_scal(h2));
      if (std::isfinite(h2[0])) cost = h2[0];
      cur_is_exact = false;
      return PXR_OK;
    };
    if (!ok) {   // HandleInvalidStep
      if (++invalid >= opt->max_consecutive_invalid_steps) { sum->termination = PXR_TERM_FAILURE; break; }
      radius *= 0.5; reuse_diag = true;
      if (user_stop(notify(sum->iterations, false, false, cost, 0.0, 0.0, radius, step_norm))) break;
      continue;
    }
    invalid = 0;
    if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) { sum->termination = PXR_TERM_CONVERGENCE; break; }
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= opt->function_tolerance * cost) { sum->termination = PXR_TERM_CONVERGENCE; break; }
    const double rel = cost_change / model_cost_change;
    if (inner_useful || rel > opt->min_relative_decrease) {   // IsStepSuccessful + HandleSuccessfulStep
      {
        Copy4 c4{{q1.p, t1.p, k1.p, X1.p}, {cur_q, cur_t, cur_k, cur_X},
                 {(int64_t)4 * n_img, (int64_t)3 * n_img, (int64_t)PXR_KPAD * n_cam, (int64_t)3 * n_pts}};
        hipLaunchKernelGGL(k_copy4, dim3((unsigned)std::min<int64_t>(1024, nblk(3 * n_pts + 1))), dim3(256), 0, st, c4);
      }
      std::swap(rec_cur, rec_cand);
      cost = cand_cost;
      cur_is_exact = false;
      if (det_fixed) { lin_md = std::max(h_lin_stats[0], 1e-300); lin_scale = det_scale_for(lin_md, cost); }   // checked with the next iteration's scalars
      RC(linearize(rec_cur));
      phase(6);
      ++sum->num_successful;
      const double tmp = 2.0 * rel - 1.0;
      radius = radius / std::max(1.0 / 3.0, 1.0 - tmp * tmp * tmp);
      radius = std::min(opt->max_radius, radius);
      decrease_factor = 2.0; reuse_diag = false;
      if (opt->gradient_tolerance > 0.0) {
        bool below = false;
        RC(gradient_below_tolerance(&below));
        if (below) { sum->termination = PXR_TERM_CONVERGENCE; break; }
      }
      if (user_stop(notify(sum->iterations, true, true, cost, cost_change, rel, radius, step_norm))) break;
    } else {   // StepRejected
      RC(reevaluate_current());
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diag = true;
      if (user_stop(notify(sum->iterations, true, false, cost, cost_change, rel, radius, step_norm))) break;
    }
  }
  PXR_HIP(hipStreamSynchronize(st));
  if (phase_timing)
    fprintf(stderr, "[pxr_ba_solve] phase ms over %d iterations: schur-build %.2f allreduce %.2f potrf %.2f potrs %.2f "
            "backsub+update %.2f eval+cost %.2f linearize %.2f other %.2f\n", sum->iterations, ph_ms[0], ph_ms[1],
            ph_ms[2], ph_ms[3], ph_ms[4], ph_ms[5], ph_ms[6], ph_ms[7]);
  sum->final_cost = cost; sum->final_radius = radius;
  sum->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_loop0).count();
  return PXR_OK;
}

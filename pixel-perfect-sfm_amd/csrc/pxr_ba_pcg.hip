// pxr_ba_pcg.hip -- iterative solve of the reduced camera system for scenes with more than
// PXR_MAX_IMAGES_DIRECT images: what ceres::ITERATIVE_SCHUR + SCHUR_JACOBI does in the reference
// (bundle_adjustment/src/bundle_optimizer.h:180-191, max_linear_solver_iterations = 200,
// bundle_adjustment_options.h:55).  [upstream Ceres 2.1: implicit_schur_complement.cc,
// conjugate_gradients_solver.cc, schur_jacobi_preconditioner.cc, levenberg_marquardt_strategy.cc.]
//
// The Schur complement S = U + D_c - sum_p W_p T_p W_p^T is never formed.  With the per-observation blocks
// W_i = B_i^T M~_i E_i (dc x 3) and T_p = (V_p + D_p)^-1 of the current linearisation (pxr_ba_solve.hip):
//     S v = sum_img U_img v_img + D_c v  -  sum_i W_i T_p(i) sum_{j in p(i)} W_j^T v_j
// is two passes over W (one point-major, one image-major: 2 x 192 B per observation at DC = 8) plus a
// block-diagonal product; U is kept as one dc x dc block per image (n_images x DC x DC doubles instead of the
// n_c x n_c matrix of the direct path: 8.2 GB at 4000 cameras).  Preconditioner: the block diagonal of S
// over the pose block of every image and the intrinsics block of every camera -- one joint block where a
// camera belongs to a single image -- with the point contributions of the block's own observations
// (a point seen twice by one block adds a cross term the reference's SCHUR_JACOBI has and this one drops).
// Conjugate gradients with Ceres' termination: Q-based, i (Q_i - Q_{i-1}) / Q_i < eta.
//
// Multi-GPU: points are sharded, so every rank owns complete T_p and a partial S v; per CG
// iteration ONE all-reduce of a camera-sized vector (n_c doubles), per linear solve one of the preconditioner
// blocks and one of the right-hand side.  All CG vectors are replicated.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pxr_ba_pcg.h"
#include "pxr_internal.h"

namespace pxr {

static inline unsigned nblk(int64_t n, int b = 256) { return (unsigned)((n + b - 1) / b); }

// device-resident control block of the conjugate-gradient loop (doubles, right behind the eight scalars of `cgs`): the loop's
// stopping decisions are taken ON the device (k_cg_decide) and every kernel of an iteration returns at once when STOP is
// set, so the host enqueues iterations without waiting for any of them and looks at the block only every few iterations
enum { CTL_STOP = 0, CTL_ITERS = 1, CTL_Q0 = 2, CTL_XR = 3, CTL_TOL_R = 4, CTL_ETA = 5, CTL_NORM_B = 6, CTL_SIZE = 8 };
#define RC(call) do { int _rc = (call); if (_rc != PXR_OK) return _rc; } while (0)

// ---- u_p = T_p (g_p + sum_j W_j^T v_j): the point-major pass ---------------------------------------------
// G lanes share one POINT: lane a walks row a of the W block of each of the point's observations (one coalesced 24 G-byte
// read per observation and group -- a thread per point read 24 doubles at a stride of the whole track: 125 us per pass at
// 1M observations against 50 us for the same bytes in k_img_wu), the three partial sums are reduced over the group with
// DPP-free shuffles in a fixed order (deterministic), lane 0 applies T_p.
template <int G>
__global__ __launch_bounds__(256) void k_pt_u(const SolveDev d, const int64_t* __restrict__ pt_ptr,
                                              const int* __restrict__ pj, const int4* __restrict__ pcols,
                                              const double* __restrict__ W, const double* __restrict__ T,
                                              const double* __restrict__ vec /* [n_c] or NULL */,
                                              const double* __restrict__ gp /* [n_points][3] or NULL */,
                                              double* __restrict__ u, const double* __restrict__ ctl /* or NULL */) {
  if (ctl && ctl[CTL_STOP] != 0.0) return;
  const int lane_a = threadIdx.x % G;
  const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const bool live = p < d.v.n_points;                      // whole groups are live or not: the shuffles below stay inside a group
  double v0 = 0.0, v1 = 0.0, v2 = 0.0;
  const bool var = live && d.pt_var[p];
  if (var && vec) {
    for (int64_t o = pt_ptr[p]; o < pt_ptr[p + 1]; ++o) {
      const int4 ci = pcols[o];                       // {pose_off, pose_dim, intr_off, intr_dim}
      if (lane_a < ci.y + ci.w) {
        const double* Wi = W + ((size_t)pj[o] * d.DC + lane_a) * 3;
        const double x = vec[lane_a < ci.y ? ci.x + lane_a : ci.z + (lane_a - ci.y)];
        v0 += Wi[0] * x; v1 += Wi[1] * x; v2 += Wi[2] * x;
      }
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) {              // fixed butterfly inside the G-lane group
    v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off); v2 += __shfl_xor(v2, off);
  }
  if (!live || lane_a != 0) return;
  if (var) {
    if (gp) { v0 += gp[3 * p]; v1 += gp[3 * p + 1]; v2 += gp[3 * p + 2]; }
    const double* Tp = T + 6 * p;
    const double t0 = Tp[0] * v0 + Tp[1] * v1 + Tp[2] * v2;
    const double t1 = Tp[1] * v0 + Tp[3] * v1 + Tp[4] * v2;
    const double t2 = Tp[2] * v0 + Tp[4] * v1 + Tp[5] * v2;
    v0 = t0; v1 = t1; v2 = t2;
  }
  u[3 * p] = v0; u[3 * p + 1] = v1; u[3 * p + 2] = v2;
}

// ---- out[cols(img)] += sign * sum_{i in chunk} W_i u_p(i): the image-major pass ----------------------------
// G lanes share one observation (lane a = row a of W_i, one coalesced 24 G-byte read per group), a workgroup
// walks one chunk of an image's observations and reduces over its groups in LDS: one atomic per (chunk, row).
template <int G>
__global__ __launch_bounds__(256) void k_img_wu(const SolveDev d, const ImgChunk* __restrict__ chunks,
                                                const int4* __restrict__ so, const double* __restrict__ W,
                                                const double* __restrict__ u, double sign, double* __restrict__ out,
                                                const double* __restrict__ ctl /* or NULL */,
                                                double* __restrict__ part /* deterministic mode: [n_chunks][DC], no atomics */) {
  __shared__ double red[256];
  if (ctl && ctl[CTL_STOP] != 0.0) return;
  const ImgChunk ch = chunks[blockIdx.x];
  const int img = ch.img, cam = d.v.d_image_camera[img];
  const int dci = d.pose_dim[img] + d.intr_dim[cam];
  if (dci == 0) return;
  const int lane_b = threadIdx.x % G, grp = threadIdx.x / G;
  constexpr int n_grp = 256 / G;
  double acc = 0.0;
  for (int64_t o = ch.begin + grp; o < ch.end; o += n_grp) {
    const int4 s = so[o];                                   // {obs, point, first partner, partners (0 = constant point)}
    if (s.w == 0 || lane_b >= dci) continue;
    const double* Wi = W + ((size_t)s.x * d.DC + lane_b) * 3;
    const double* up = u + 3 * (size_t)s.y;
    acc += Wi[0] * up[0] + Wi[1] * up[1] + Wi[2] * up[2];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < dci) {
    double t = 0.0;
    for (int g = 0; g < n_grp; ++g) t += red[g * G + threadIdx.x];
    if (part) part[(size_t)blockIdx.x * d.DC + threadIdx.x] = t;             // (summed per column, in order, by k_col_finish)
    else if (t != 0.0) atomicAdd(out + col_index(d, img, cam, threadIdx.x), sign * t);
  }
}

// ---- deterministic mode: out[c] = sum over the column's entries (image, a), images ascending, of
//        [WITH_U: (U_img vec)_a]  +  sign * sum over the image's chunks, in order, of part[chunk][a]
// One wavefront per column: lane l takes entries l, l + 64, ... (a camera shared by thousands of images has thousands of entries
// in its intrinsics columns), then a fixed butterfly -- the same order on every run, no atomics.
template <bool WITH_U>
__global__ __launch_bounds__(256) void k_col_finish(const SolveDev d, const int* __restrict__ col_ent_ptr, const int2* __restrict__ col_ent,
                                                    const int* __restrict__ chunk_ptr, const double* __restrict__ part, double sign,
                                                    const double* __restrict__ Ublk, const double* __restrict__ vec,
                                                    double* __restrict__ out, const double* __restrict__ ctl) {
  if (ctl && ctl[CTL_STOP] != 0.0) return;
  const int c = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (c >= d.n_c) return;                                   // (whole wavefronts)
  double acc = 0.0;
  for (int e = col_ent_ptr[c] + lane; e < col_ent_ptr[c + 1]; e += 64) {
    const int2 en = col_ent[e];
    const int img = en.x, a = en.y, cam = d.v.d_image_camera[img];
    double t = 0.0;
    if (WITH_U) {
      const int dci = d.pose_dim[img] + d.intr_dim[cam];
      const double* Ua = Ublk + ((size_t)img * d.DC + a) * d.DC;
      for (int b = 0; b < dci; ++b) t += Ua[b] * vec[col_index(d, img, cam, b)];
    }
    for (int k = chunk_ptr[img]; k < chunk_ptr[img + 1]; ++k) t += sign * part[(size_t)k * d.DC + a];
    acc += t;
  }
  acc = wave_sum(acc);
  if (lane == 0) out[c] = acc;
}

// ---- out += U v, U = one dc x dc block per image ------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ublk_matvec(const SolveDev d, const double* __restrict__ Ublk,
                                                     const double* __restrict__ vec, double* __restrict__ out,
                                                     const double* __restrict__ ctl) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int img = (int)(t / d.DC), a = (int)(t - (int64_t)img * d.DC);
  if (img >= d.v.n_images || ctl[CTL_STOP] != 0.0) return;
  const int cam = d.v.d_image_camera[img];
  const int dci = d.pose_dim[img] + d.intr_dim[cam];
  if (a >= dci) return;
  const double* Ua = Ublk + ((size_t)img * d.DC + a) * d.DC;
  double acc = 0.0;
  for (int b = 0; b < dci; ++b) acc += Ua[b] * vec[col_index(d, img, cam, b)];
  atomicAdd(out + col_index(d, img, cam, a), acc);
}

// ---- Mloc[img] -= sum_{i in chunk} (W_i T_p) W_i^T: the point part of an image's own diagonal block -------
template <int G>
__global__ __launch_bounds__(256) void k_img_block(const SolveDev d, const ImgChunk* __restrict__ chunks,
                                                   const int4* __restrict__ so, const double* __restrict__ W,
                                                   const double* __restrict__ T, double* __restrict__ Mloc,
                                                   double* __restrict__ mpart /* deterministic mode: [n_chunks][DC][DC], no atomics */) {
  extern __shared__ double red[];                            // [256 / G][G][G]
  const ImgChunk ch = chunks[blockIdx.x];
  const int img = ch.img, cam = d.v.d_image_camera[img];
  const int dci = d.pose_dim[img] + d.intr_dim[cam];
  if (dci == 0) return;
  const int lane_b = threadIdx.x % G, grp = threadIdx.x / G;
  constexpr int n_grp = 256 / G;
  double acc[G];
#pragma unroll
  for (int b = 0; b < G; ++b) acc[b] = 0.0;
  const int64_t n_it = (ch.end - ch.begin + n_grp - 1) / n_grp;                // uniform trip count: shuffles need every lane
  for (int64_t it = 0; it < n_it; ++it) {
    const int64_t o = ch.begin + it * n_grp + grp;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, y0 = 0.0, y1 = 0.0, y2 = 0.0;
    if (o < ch.end) {
      const int4 s = so[o];
      if (s.w != 0 && lane_b < dci) {
        const double* Wi = W + ((size_t)s.x * d.DC + lane_b) * 3;
        w0 = Wi[0]; w1 = Wi[1]; w2 = Wi[2];
        const double* Tp = T + 6 * (size_t)s.y;
        y0 = w0 * Tp[0] + w1 * Tp[1] + w2 * Tp[2];
        y1 = w0 * Tp[1] + w1 * Tp[3] + w2 * Tp[4];
        y2 = w0 * Tp[2] + w1 * Tp[4] + w2 * Tp[5];
      }
    }
#pragma unroll
    for (int b = 0; b < G; ++b) {
      const int src = (threadIdx.x & 63) - lane_b + b;     // lane b of this group (G divides 64)
      acc[b] += y0 * __shfl(w0, src) + y1 * __shfl(w1, src) + y2 * __shfl(w2, src);
    }
  }
#pragma unroll
  for (int b = 0; b < G; ++b) red[((size_t)grp * G + lane_b) * G + b] = acc[b];
  __syncthreads();
  for (int e = threadIdx.x; e < dci * dci; e += blockDim.x) {
    const int a = e / dci, b = e - a * dci;
    double t = 0.0;
    for (int g = 0; g < n_grp; ++g) t += red[((size_t)g * G + a) * G + b];
    if (mpart) mpart[((size_t)blockIdx.x * d.DC + a) * d.DC + b] = t;
    else if (t != 0.0) atomicAdd(Mloc + ((size_t)img * d.DC + a) * d.DC + b, -t);
  }
}

// deterministic mode: Mloc[img] = U_img - the image's chunks' parts, in order
__global__ __launch_bounds__(256) void k_mloc_finish(const SolveDev d, const int* __restrict__ chunk_ptr, const double* __restrict__ mpart,
                                                     const double* __restrict__ Ublk, double* __restrict__ Mloc) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int DC = d.DC;
  const int img = (int)(t / (DC * DC)), e = (int)(t - (int64_t)img * DC * DC);
  if (img >= d.v.n_images) return;
  const int a = e / DC, b = e - a * DC;
  const int dci = d.pose_dim[img] + d.intr_dim[d.v.d_image_camera[img]];
  double v = Ublk[(size_t)img * DC * DC + e];
  if (a < dci && b < dci)
    for (int k = chunk_ptr[img]; k < chunk_ptr[img + 1]; ++k) v -= mpart[((size_t)k * DC + a) * DC + b];
  Mloc[(size_t)img * DC * DC + e] = v;
}

// ---- preconditioner blocks --------------------------------------------------------------------------------
// every reduced-system column belongs to exactly one block (col_group[c] = {block, position})
__global__ __launch_bounds__(256) void k_pre_assemble(const SolveDev d, const int2* __restrict__ col_group,
                                                      const double* __restrict__ Mloc, double* __restrict__ Gm) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int DC = d.DC;
  const int img = (int)(t / (DC * DC)), e = (int)(t - (int64_t)img * DC * DC);
  if (img >= d.v.n_images) return;
  const int a = e / DC, b = e - a * DC;
  const int cam = d.v.d_image_camera[img];
  const int dci = d.pose_dim[img] + d.intr_dim[cam];
  if (a >= dci || b >= dci) return;
  const int2 ga = col_group[col_index(d, img, cam, a)], gb = col_group[col_index(d, img, cam, b)];
  if (ga.x != gb.x) return;
  const double v = Mloc[((size_t)img * DC + a) * DC + b];
  if (v != 0.0) atomicAdd(Gm + ((size_t)ga.x * PCG_GS + ga.y) * PCG_GS + gb.y, v);
}

// deterministic mode: entry (ya, yb) of block g = the ordered sum over the images that hold both columns (one image for a pose /
// joint block, every image of the camera for a shared intrinsics block)
__global__ __launch_bounds__(256) void k_pre_assemble_det(const SolveDev d, int n_groups, const int* __restrict__ group_size,
                                                          const int* __restrict__ group_cols, const int* __restrict__ col_ent_ptr,
                                                          const int2* __restrict__ col_ent, const double* __restrict__ Mloc,
                                                          double* __restrict__ Gm) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int g = (int)(t / (PCG_GS * PCG_GS)), e = (int)(t - (int64_t)g * PCG_GS * PCG_GS);
  if (g >= n_groups) return;
  const int ya = e / PCG_GS, yb = e - ya * PCG_GS, ng = group_size[g];
  double acc = 0.0;
  if (ya < ng && yb < ng) {
    const int ca = group_cols[g * PCG_GS + ya], cb = group_cols[g * PCG_GS + yb];
    for (int k = col_ent_ptr[ca]; k < col_ent_ptr[ca + 1]; ++k) {
      const int2 en = col_ent[k];
      const int img = en.x, cam = d.v.d_image_camera[img], pd = d.pose_dim[img];
      int b = -1;
      if (cb >= d.pose_off[img] && cb < d.pose_off[img] + pd) b = cb - d.pose_off[img];
      else if (cb >= d.intr_off[cam] && cb < d.intr_off[cam] + d.intr_dim[cam]) b = pd + (cb - d.intr_off[cam]);
      if (b >= 0) acc += Mloc[((size_t)img * d.DC + en.y) * d.DC + b];
    }
  }
  Gm[(size_t)g * PCG_GS * PCG_GS + e] = acc;
}

// One wavefront per block: Gauss-Jordan inversion of the SPD block + damping, lane = row of [A | I].
__global__ __launch_bounds__(256) void k_pre_invert(int n_groups, const int* __restrict__ group_size,
                                                    const int* __restrict__ group_cols, const double* __restrict__ damp,
                                                    double inv_radius, double* __restrict__ Gm, int* __restrict__ fail) {
  const int g = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (g >= n_groups) return;
  const int ng = group_size[g];
  double A[PCG_GS], Iv[PCG_GS];
  double* Gg = Gm + (size_t)g * PCG_GS * PCG_GS;
#pragma unroll
  for (int j = 0; j < PCG_GS; ++j) {
    A[j] = (lane < ng && j < ng) ? Gg[lane * PCG_GS + j] : 0.0;
    Iv[j] = (j == lane) ? 1.0 : 0.0;
  }
  if (lane < ng) {
    const double dd = damp[group_cols[g * PCG_GS + lane]] * inv_radius;
#pragma unroll
    for (int j = 0; j < PCG_GS; ++j) if (j == lane) A[j] += dd;
  }
  bool bad = false;
#pragma unroll
  for (int k = 0; k < PCG_GS; ++k) {
    if (k < ng) {                                              // uniform over the wavefront
      const double pivot = __shfl(A[k], k);
      if (!(pivot > 0.0) || !isfinite(pivot)) bad = true;
      const double ipiv = 1.0 / pivot;
      const double f = (lane == k) ? 0.0 : A[k] * ipiv;         // this row's multiplier, taken before column k changes
#pragma unroll
      for (int j = 0; j < PCG_GS; ++j) {
        const double rk = __shfl(A[j], k), ri = __shfl(Iv[j], k);   // pivot row, read before lane k rescales it
        if (lane == k) { A[j] *= ipiv; Iv[j] *= ipiv; }
        else { A[j] -= f * rk; Iv[j] -= f * ri; }
      }
    }
  }
  if (bad && lane == 0) atomicExch(fail, 1);
  if (lane < ng) {
#pragma unroll
    for (int j = 0; j < PCG_GS; ++j) if (j < ng) Gg[lane * PCG_GS + j] = Iv[j];
  }
}

// ---- conjugate-gradient vector kernels; device scalars cgs = {rho, rho_prev, p.q, x.b, x.r, r.r, |b|^2, -} -------
// The dot products are DETERMINISTIC: every 256-thread block reduces in a fixed tree and writes its partial sum to
// part[slot][block]; k_cg_finalize adds the partials in block order.  All CG vectors are replicated over the ranks and
// built from all-reduced (hence bit-identical) data, so with order-fixed reductions alpha, beta and every stopping
// decision of the host loop are bit-identical on all ranks -- a rank leaving the loop one iteration early would leave
// the others waiting in the next all-reduce.  (Floating-point atomics would make the last bits rank-dependent.)
constexpr int CG_SLOTS = 8;

__device__ __forceinline__ void block_partial(double v, double* __restrict__ part, int slot, int nb) {   // 256 threads
  __shared__ double sh[CG_SLOTS][4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[slot][threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)slot * nb + blockIdx.x] = (sh[slot][0] + sh[slot][1]) + (sh[slot][2] + sh[slot][3]);
}
// the sum of a slot's partials, formed by every block of a consumer kernel for itself instead of waiting for a one-thread
// "finalize" launch: lane l of the block's first wavefront adds partials l, l + 64, ... in order, then a fixed butterfly --
// the same order in every block and on every rank, hence the same bits everywhere.  All threads of the block must call it.
__device__ __forceinline__ double slot_sum(const double* __restrict__ part, int slot, int nb) {
  __shared__ double total[CG_SLOTS];
  if (threadIdx.x < 64) {
    double t = 0.0;
    for (int b = threadIdx.x; b < nb; b += 64) t += part[(size_t)slot * nb + b];
    t = wave_sum(t);
    if (threadIdx.x == 0) total[slot] = t;
  }
  __syncthreads();
  return total[slot];
}

__global__ void k_cg_finalize(const double* __restrict__ part, int nb, unsigned slot_mask, double* __restrict__ cgs) {   // 64 threads
  for (int s = 0; s < CG_SLOTS; ++s)
    if ((slot_mask >> s) & 1u) { const double t = slot_sum(part, s, nb); if (threadIdx.x == 0) cgs[s] = t; }
}

// set-up of the loop's control block: |b|, the tolerances, "not stopped"; |b| = 0 or not finite stops before the first iteration
__global__ void k_cg_init(double* __restrict__ cgs, double* __restrict__ ctl, double r_tolerance, double eta, const int* __restrict__ fail) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double norm_b = sqrt(cgs[6]);
  ctl[CTL_NORM_B] = norm_b;
  ctl[CTL_TOL_R] = r_tolerance > 0.0 ? r_tolerance * norm_b : -1.0;
  ctl[CTL_ETA] = eta;
  ctl[CTL_ITERS] = 0.0; ctl[CTL_Q0] = 0.0; ctl[CTL_XR] = 0.0;
  ctl[CTL_STOP] = (*fail != 0 || !isfinite(norm_b) || norm_b == 0.0) ? 1.0 : 0.0;
  cgs[0] = 0.0; cgs[1] = 0.0;
}

// z = M^-1 r, partials of rho = r.z
__global__ __launch_bounds__(256) void k_pre_apply(int n, const int2* __restrict__ col_group,
                                                   const int* __restrict__ group_size, const int* __restrict__ group_cols,
                                                   const double* __restrict__ Ginv, const double* __restrict__ r,
                                                   double* __restrict__ z, double* __restrict__ part, const double* __restrict__ ctl) {
  if (ctl[CTL_STOP] != 0.0) return;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (c < n) {
    const int2 g = col_group[c];
    const int ng = group_size[g.x];
    const double* row = Ginv + ((size_t)g.x * PCG_GS + g.y) * PCG_GS;
    const int* cols = group_cols + (size_t)g.x * PCG_GS;
    double acc = 0.0;
    for (int b = 0; b < ng; ++b) acc += row[b] * r[cols[b]];
    z[c] = acc;
    v = r[c] * acc;
  }
  block_partial(v, part, 0, gridDim.x);
}

// rho from its partials; p = z + (rho / rho_prev) p; q = 0 for the products that follow
__global__ __launch_bounds__(256) void k_cg_p_update(int n, int first, const double* __restrict__ z, double* __restrict__ p,
                                                     double* __restrict__ q, const double* __restrict__ part, int nb,
                                                     double* __restrict__ cgs, const double* __restrict__ ctl) {
  if (ctl[CTL_STOP] != 0.0) return;
  const double rho = slot_sum(part, 0, nb);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double beta = first ? 0.0 : rho / cgs[1];
    p[i] = first ? z[i] : z[i] + beta * p[i];
    q[i] = 0.0;
  }
  if (i == 0) cgs[0] = rho;
}

// q += D p / radius (after the all-reduce of the partial products), partials of p.q
__global__ __launch_bounds__(256) void k_cg_q_finish(int n, const double* __restrict__ damp, double inv_radius,
                                                     const double* __restrict__ p, double* __restrict__ q,
                                                     double* __restrict__ part, const double* __restrict__ ctl) {
  if (ctl[CTL_STOP] != 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = 0.0;
  if (i < n) {
    const double qi = q[i] + damp[i] * inv_radius * p[i];
    q[i] = qi;
    v = p[i] * qi;
  }
  block_partial(v, part, 2, gridDim.x);
}

// p.q from its partials; x += alpha p, r -= alpha q unless the matrix turned out indefinite along p (p.q <= 0: the loop
// stops and keeps x); partials of x.b, x.r, r.r
__global__ __launch_bounds__(256) void k_cg_xr_update(int n, const double* __restrict__ p, const double* __restrict__ q,
                                                      const double* __restrict__ b, double* __restrict__ x,
                                                      double* __restrict__ r, double* __restrict__ cgs,
                                                      double* __restrict__ part, int nb, const double* __restrict__ ctl) {
  if (ctl[CTL_STOP] != 0.0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double pq = slot_sum(part, 2, nb), rho = cgs[0];
  const bool ok = pq > 0.0 && isfinite(pq) && rho != 0.0 && isfinite(rho);
  double xb = 0.0, xr = 0.0, rr = 0.0;
  if (i < n) {
    double xi = x[i], ri = r[i];
    if (ok) {
      const double alpha = rho / pq;
      xi += alpha * p[i]; ri -= alpha * q[i];
      x[i] = xi; r[i] = ri;
    }
    xb = xi * b[i]; xr = xi * ri; rr = ri * ri;
  }
  if (i == 0) cgs[2] = pq;
  block_partial(xb, part, 3, gridDim.x);
  block_partial(xr, part, 4, gridDim.x);
  block_partial(rr, part, 5, gridDim.x);
}

// the end of iteration `it`: Ceres' tests ([upstream] conjugate_gradients_solver.cc) on the device
__global__ void k_cg_decide(int it, const double* __restrict__ part, int nb, double* __restrict__ cgs, double* __restrict__ ctl) {
  if (blockIdx.x != 0 || ctl[CTL_STOP] != 0.0) return;                 // one block of 64 threads
  const double xb = slot_sum(part, 3, nb), xr = slot_sum(part, 4, nb), rr = slot_sum(part, 5, nb);
  if (threadIdx.x != 0) return;
  const double rho = cgs[0], pq = cgs[2];
  cgs[3] = xb; cgs[4] = xr; cgs[5] = rr;
  cgs[1] = rho;                                                    // rho_prev of the next iteration
  if (!isfinite(rho) || rho == 0.0) { ctl[CTL_STOP] = 1.0; return; }   // "Numerical failure. rho = r'z = 0": iteration not counted
  if (!(pq > 0.0) || !isfinite(pq)) { ctl[CTL_STOP] = 1.0; return; }   // "Matrix is indefinite": the last x is kept
  ctl[CTL_ITERS] = (double)it;
  ctl[CTL_XR] = xr;
  // Q = x.Sx / 2 - x.b = -(x.b + x.r) / 2 with r = b - S x
  const double Q1 = -0.5 * (xb + xr);
  const double zeta = it * (Q1 - ctl[CTL_Q0]) / Q1;
  if (zeta < ctl[CTL_ETA]) { ctl[CTL_STOP] = 1.0; return; }
  ctl[CTL_Q0] = Q1;
  if (ctl[CTL_TOL_R] > 0.0 && sqrt(rr) <= ctl[CTL_TOL_R]) ctl[CTL_STOP] = 1.0;
}

__global__ void k_vec_add(int n, const double* __restrict__ a, double* __restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a[i];
}

__global__ __launch_bounds__(256) void k_dot(int n, const double* __restrict__ a, const double* __restrict__ b,
                                             double* __restrict__ part, int slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  block_partial(i < n ? a[i] * b[i] : 0.0, part, slot, gridDim.x);
}

// ---- host driver ------------------------------------------------------------------------------------------------
template <typename F>
static void launch_by_g(int DC, F&& f) { if (DC <= 8) f(std::integral_constant<int, 8>()); else if (DC <= 16) f(std::integral_constant<int, 16>()); else f(std::integral_constant<int, 32>()); }

int pcg_solve(PcgArgs& a, double inv_radius, const pxr_lm_options* opt, const std::function<int(double*, int64_t)>& ar,
              PcgResult* res) {
  hipStream_t st = a.st;
  const SolveDev& d = a.dv;
  const int n = d.n_c, DC = d.DC;
  const int n_img = d.v.n_images;
  const int64_t n_pts = d.v.n_points;
  res->iterations = 0; res->ok = false; res->x_dot_r = 0.0;
  const bool verbose = std::getenv("PXR_VERBOSE") != nullptr;

  // ---- right-hand side b = g_c - sum_i W_i T_p g_p (g_c is already the sum over the ranks) --------------
  // Deterministic mode (a.det): every sum that floating-point atomics would form in arrival order -- a chunk's part of W u or of an
  // image's block into the image's slots, several images' parts into the columns of a shared camera -- is formed from per-chunk
  // partials in a fixed order instead (k_col_finish, k_mloc_finish, k_pre_assemble_det): the same bits on every run.
  const dim3 col_grid(nblk((int64_t)n * 64));
  PXR_HIP(hipMemsetAsync(a.b, 0, sizeof(double) * n, st));
  launch_by_g(DC, [&](auto G) {
    constexpr int GG = decltype(G)::value;
    hipLaunchKernelGGL(k_pt_u<GG>, dim3(nblk(n_pts * GG)), dim3(256), 0, st, d, a.pt_ptr, a.part_obs, a.obs_cols, a.W, a.T,
                       (const double*)nullptr, a.gp, a.u, (const double*)nullptr);
  });
  if (a.n_chunks > 0)
    launch_by_g(DC, [&](auto G) {
      hipLaunchKernelGGL(k_img_wu<decltype(G)::value>, dim3((unsigned)a.n_chunks), dim3(256), 0, st, d, a.chunks, a.so, a.W, a.u, -1.0, a.b,
                         (const double*)nullptr, a.det ? a.wpart : (double*)nullptr);
    });
  if (a.det)
    hipLaunchKernelGGL(k_col_finish<false>, col_grid, dim3(256), 0, st, d, a.col_ent_ptr, a.col_ent, a.chunk_ptr, (const double*)a.wpart, -1.0,
                       (const double*)nullptr, (const double*)nullptr, a.b, (const double*)nullptr);
  // ---- preconditioner: local blocks = U_img - sum_i Y_i W_i^T, gathered into the column groups ----------
  if (!a.det) PXR_HIP(hipMemcpyAsync(a.Mloc, a.Ublk, sizeof(double) * (size_t)n_img * DC * DC, hipMemcpyDeviceToDevice, st));
  if (a.n_chunks > 0)
    launch_by_g(DC, [&](auto G) {
      constexpr int GG = decltype(G)::value;
      hipLaunchKernelGGL(k_img_block<GG>, dim3((unsigned)a.n_chunks), dim3(256), sizeof(double) * 256 * GG, st, d, a.chunks, a.so, a.W, a.T, a.Mloc,
                         a.det ? a.mpart : (double*)nullptr);
    });
  if (a.det) {
    hipLaunchKernelGGL(k_mloc_finish, dim3(nblk((int64_t)n_img * DC * DC)), dim3(256), 0, st, d, a.chunk_ptr, (const double*)a.mpart, a.Ublk, a.Mloc);
    hipLaunchKernelGGL(k_pre_assemble_det, dim3(nblk((int64_t)a.n_groups * PCG_GS * PCG_GS)), dim3(256), 0, st, d, a.n_groups, a.group_size,
                       a.group_cols, a.col_ent_ptr, a.col_ent, (const double*)a.Mloc, a.Gm);
  } else {
    PXR_HIP(hipMemsetAsync(a.Gm, 0, sizeof(double) * (size_t)a.n_groups * PCG_GS * PCG_GS, st));
    hipLaunchKernelGGL(k_pre_assemble, dim3(nblk((int64_t)n_img * DC * DC)), dim3(256), 0, st, d, a.col_group, a.Mloc, a.Gm);
  }
  RC(hip_check(hipGetLastError(), "pcg set-up kernels"));
  RC(ar(a.b, n));
  RC(ar(a.Gm, (int64_t)a.n_groups * PCG_GS * PCG_GS));
  hipLaunchKernelGGL(k_vec_add, dim3(nblk(n)), dim3(256), 0, st, n, a.gc, a.b);
  PXR_HIP(hipMemsetAsync(a.d_fail, 0, sizeof(int), st));
  hipLaunchKernelGGL(k_pre_invert, dim3(nblk((int64_t)a.n_groups * 64)), dim3(256), 0, st, a.n_groups, a.group_size, a.group_cols,
                     a.damp_c, inv_radius, a.Gm, a.d_fail);
  // ---- x = 0, r = b ------------------------------------------------------------------------------------------
  PXR_HIP(hipMemsetAsync(a.x, 0, sizeof(double) * n, st));
  PXR_HIP(hipMemcpyAsync(a.r, a.b, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  PXR_HIP(hipMemsetAsync(a.cgs, 0, sizeof(double) * (8 + CTL_SIZE), st));
  const int nb = (int)nblk(n);
  double* ctl = a.cgs + 8;
  hipLaunchKernelGGL(k_dot, dim3(nb), dim3(256), 0, st, n, a.b, a.b, a.cg_part, 6);
  hipLaunchKernelGGL(k_cg_finalize, dim3(1), dim3(64), 0, st, a.cg_part, nb, 1u << 6, a.cgs);
  hipLaunchKernelGGL(k_cg_init, dim3(1), dim3(64), 0, st, a.cgs, ctl, opt->linear_r_tolerance, opt->eta, a.d_fail);
  // ---- the loop: no host round trip inside an iteration.  The control block comes to the host after iteration 2 (where
  // Ceres' Q test usually ends a well-preconditioned solve) and then after every fourth; iterations enqueued past the
  // stop are a handful of kernels that return at their first instruction.
  const int max_it = opt->max_linear_solver_iterations > 0 ? opt->max_linear_solver_iterations : 200;
  struct { double cgs[8]; double ctl[CTL_SIZE]; } h;
  int h_fail = 0;
  for (int it = 1; it <= max_it; ++it) {
    hipLaunchKernelGGL(k_pre_apply, dim3(nb), dim3(256), 0, st, n, a.col_group, a.group_size, a.group_cols, a.Gm, a.r, a.z, a.cg_part, ctl);
    hipLaunchKernelGGL(k_cg_p_update, dim3(nb), dim3(256), 0, st, n, it == 1 ? 1 : 0, a.z, a.p, a.q, a.cg_part, nb, a.cgs, ctl);
    // q = S p (partial over this rank's points)
    if (!a.det) hipLaunchKernelGGL(k_ublk_matvec, dim3(nblk((int64_t)n_img * DC)), dim3(256), 0, st, d, a.Ublk, a.p, a.q, ctl);
    launch_by_g(DC, [&](auto G) {
      constexpr int GG = decltype(G)::value;
      hipLaunchKernelGGL(k_pt_u<GG>, dim3(nblk(n_pts * GG)), dim3(256), 0, st, d, a.pt_ptr, a.part_obs, a.obs_cols, a.W, a.T, a.p,
                         (const double*)nullptr, a.u, ctl);
    });
    if (a.n_chunks > 0)
      launch_by_g(DC, [&](auto G) {
        hipLaunchKernelGGL(k_img_wu<decltype(G)::value>, dim3((unsigned)a.n_chunks), dim3(256), 0, st, d, a.chunks, a.so, a.W, a.u, -1.0, a.q, ctl,
                           a.det ? a.wpart : (double*)nullptr);
      });
    if (a.det)          // U p and the chunks' parts of -W u, column by column in a fixed order
      hipLaunchKernelGGL(k_col_finish<true>, col_grid, dim3(256), 0, st, d, a.col_ent_ptr, a.col_ent, a.chunk_ptr, (const double*)a.wpart, -1.0,
                         a.Ublk, (const double*)a.p, a.q, (const double*)ctl);
    RC(hip_check(hipGetLastError(), "pcg matvec kernels"));
    RC(ar(a.q, n));                   // (past the stop: every rank reduces the same stale vector -- the ranks stay in step)
    hipLaunchKernelGGL(k_cg_q_finish, dim3(nb), dim3(256), 0, st, n, a.damp_c, inv_radius, a.p, a.q, a.cg_part, ctl);
    hipLaunchKernelGGL(k_cg_xr_update, dim3(nb), dim3(256), 0, st, n, a.p, a.q, a.b, a.x, a.r, a.cgs, a.cg_part, nb, ctl);
    hipLaunchKernelGGL(k_cg_decide, dim3(1), dim3(64), 0, st, it, a.cg_part, nb, a.cgs, ctl);
    if (verbose || it == 2 || it % 4 == 0 || it == max_it) {
      PXR_HIP(hipMemcpyAsync(&h, a.cgs, sizeof(h), hipMemcpyDeviceToHost, st));
      PXR_HIP(hipMemcpyAsync(&h_fail, a.d_fail, sizeof(int), hipMemcpyDeviceToHost, st));
      PXR_HIP(hipStreamSynchronize(st));
      if (verbose && (int)h.ctl[CTL_ITERS] == it)
        fprintf(stderr, "[pxr_ba_solve]   cg %3d |r|/|b| %.3e Q %.9e\n", it, std::sqrt(h.cgs[5]) / h.ctl[CTL_NORM_B], -0.5 * (h.cgs[3] + h.cgs[4]));
      if (h.ctl[CTL_STOP] != 0.0) break;
    }
  }
  if (h_fail) { if (verbose) fprintf(stderr, "[pxr_ba_solve] pcg: a preconditioner block is not positive definite\n"); return PXR_OK; }
  if (!std::isfinite(h.ctl[CTL_NORM_B])) return PXR_OK;
  if (h.ctl[CTL_NORM_B] == 0.0) { res->ok = true; return PXR_OK; }          // x = 0 solves it
  res->iterations = (int)h.ctl[CTL_ITERS];
  res->x_dot_r = h.ctl[CTL_XR];
  res->ok = res->iterations > 0;
  return PXR_OK;
}

// ---- linearisation helpers of the block form of U -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_diag_from_blocks(const SolveDev d, const double* __restrict__ Ublk,
                                                          double* __restrict__ diag) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int img = (int)(t / d.DC), a = (int)(t - (int64_t)img * d.DC);
  if (img >= d.v.n_images) return;
  const int cam = d.v.d_image_camera[img];
  if (a >= d.pose_dim[img] + d.intr_dim[cam]) return;
  const double v = Ublk[((size_t)img * d.DC + a) * d.DC + a];
  if (v != 0.0) atomicAdd(diag + col_index(d, img, cam, a), v);
}

// deterministic linearisation of the block form (see pxr_ba_pcg.h)
__global__ __launch_bounds__(256) void k_img_finish(const SolveDev d, const int* __restrict__ kchunk_ptr, const double* __restrict__ kpart,
                                                    int ne_max, double* __restrict__ Ublk, double* __restrict__ gimg) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int img = (int)(t / ne_max), e = (int)(t - (int64_t)img * ne_max);
  if (img >= d.v.n_images) return;
  const int dc = d.pose_dim[img] + d.intr_dim[d.v.d_image_camera[img]];
  const int NP = dc * (dc + 1) / 2;
  if (e >= NP + dc) return;
  double v = 0.0;
  for (int k = kchunk_ptr[img]; k < kchunk_ptr[img + 1]; ++k) v += kpart[(size_t)k * ne_max + e];
  if (e >= NP) { gimg[(size_t)img * d.DC + (e - NP)] = v; return; }
  int a = 0, rem = e;                                      // k_img's numbering of the upper triangle
  while (rem >= dc - a) { rem -= dc - a; ++a; }
  const int b = a + rem;
  double* Ub = Ublk + (size_t)img * d.DC * d.DC;
  Ub[a * d.DC + b] = v;
  Ub[b * d.DC + a] = v;
}
// one wavefront per column: g_c and diag(U) as the ordered sums over the column's entries
__global__ __launch_bounds__(256) void k_col_gather(const SolveDev d, const int* __restrict__ col_ent_ptr, const int2* __restrict__ col_ent,
                                                    const double* __restrict__ gimg, const double* __restrict__ Ublk,
                                                    double* __restrict__ gc, double* __restrict__ diag) {
  const int c = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (c >= d.n_c) return;
  double g = 0.0, u = 0.0;
  for (int e = col_ent_ptr[c] + lane; e < col_ent_ptr[c + 1]; e += 64) {
    const int2 en = col_ent[e];
    g += gimg[(size_t)en.x * d.DC + en.y];
    u += Ublk[((size_t)en.x * d.DC + en.y) * d.DC + en.y];
  }
  g = wave_sum(g); u = wave_sum(u);
  if (lane == 0) { gc[c] = g; diag[c] = u; }
}

int pcg_blocks_from_partials(hipStream_t st, const SolveDev& d, const int* kchunk_ptr, const double* kpart, int ne_max, double* Ublk,
                             double* gimg, const int* col_ent_ptr, const int2* col_ent, double* diag, double* gc) {
  hipLaunchKernelGGL(k_img_finish, dim3(nblk((int64_t)d.v.n_images * ne_max)), dim3(256), 0, st, d, kchunk_ptr, kpart, ne_max, Ublk, gimg);
  hipLaunchKernelGGL(k_col_gather, dim3(nblk((int64_t)d.n_c * 64)), dim3(256), 0, st, d, col_ent_ptr, col_ent, (const double*)gimg, (const double*)Ublk, gc, diag);
  return hip_check(hipGetLastError(), "deterministic block linearisation");
}

int pcg_diag_from_blocks(hipStream_t st, const SolveDev& d, const double* Ublk, double* diag) {
  hipLaunchKernelGGL(k_diag_from_blocks, dim3(nblk((int64_t)d.v.n_images * d.DC)), dim3(256), 0, st, d, Ublk, diag);
  return hip_check(hipGetLastError(), "k_diag_from_blocks");
}

}  // namespace pxr

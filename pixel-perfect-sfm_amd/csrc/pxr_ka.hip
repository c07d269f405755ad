// pxr_ka.hip -- featuremetric keypoint adjustment on gfx950.
//
// Reference path: FeatureMetricKeypointOptimizer::RunParallel -> one ceres::Problem per problem
// label, one FeatureMetric2DCostFunctor per intra-track edge (2 bicubic interpolations per
// residual block, 128 x 2 x 2 Jacobian), solved by Ceres TR-LM with box bounds on a CPU thread
// (keypoint_adjustment/src/featuremetric_keypoint_optimizer.h:69-202, keypoint_optimizer.h:77-157);
// and the localization variant with unary FeatureReference2DCostFunctor blocks
// (localization/src/query_keypoint_optimizer.h:122-172, single_query_keypoint_optimizer.h:86-170).
//
// MI355X design: ONE workgroup owns one sub-problem for the whole solve -- no host round trips,
// thousands of sub-problems in flight.  Setup finds the connected components of the sub-problem's
// variable nodes (tracks are independent, a query's keypoints are independent): the normal matrix
// is block diagonal, stored and factored per component.  Per LM iteration the workgroup
//   1. evaluates every NODE once (16 lanes per node, 8 channels per lane: normalised descriptor
//      + image-space gradients, 3 KiB per node kept in an L2-resident scratch) instead of twice
//      per EDGE as the reference does -- a track of 10 nodes with its complete match graph reads
//      each 4 KiB stencil once, not 18 times;
//   2. walks the edges / unary terms (16 lanes each): the dot products that define the
//      robustified normal block, reduced with DPP inside the row, scattered into the component's
//      dense block;
//   3. runs Ceres' trust-region step: Jacobi scaling, LM damping, Cholesky per component (one
//      thread for a 2x2 block, one wavefront for <= 64 unknowns, the workgroup beyond that),
//      projected Armijo line search along the step (bounds), step acceptance and radius update.
// [upstream Ceres 2.1] semantics restated as in oracle/pxo_solve.c; the oracle is the parity
// target (real Ceres is not available: parity unpinned w.r.t. the reference binary).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "pxr_device.h"
#include "pxr_interp.h"
#include "pxr_internal.h"

namespace pxr {

constexpr int KA_NT = 256;     // threads per sub-problem workgroup
constexpr int KA_NLDS = 84;    // LDS budget for the damped blocks: KA_NLDS^2 doubles = 55 KiB; with the 22.5 KiB of static metadata caches (sh_nodes, sh_edges, sh_sq) under 80 KiB: two workgroups per CU

struct KaArgs {
  pxr_ka_view v;
  const void* arena; const int32_t* corners; const double* scales; int H, W;
  int l2_normalize; int float_simd; int check_bounds; int lds_elems;   // lds_elems: doubles of dynamic LDS for the damped blocks
  int lds_state_n;       // > 0: the whole LM state of a sub-problem lives in LDS too (normal matrix + 8 vectors + 4 index arrays of this many unknowns)
  pxr_loss loss; double bound; pxr_lm_options opt;
  // scratch
  double* desc;          // [n_nodes][3][C]: f, df/dx, df/dy
  double* kp_cand;       // [n_nodes][2]
  int* var_of_node;      // [n_nodes] first unknown of the node inside its sub-problem, or -1
  uint8_t* used;         // [n_nodes] (zeroed by host)
  int* label;            // [n_nodes] component label = local position of the component's first node
  int* ipos;             // 4 arrays [n_nodes], indexed by position in d_prob_nodes: cnt, cstart, hoff, cidx
  int* irow;             // 3 arrays [2 n_nodes], indexed by unknown: row_off, row_v0, row_nc
  int* comp_v0;          // [n_nodes] first unknown of each component, per sub-problem at its node offset
  double* vec;           // 9 vectors of [2 * n_nodes]: g, gun, scale, diag, step, delta, lo, hi, rhs
  const int64_t* prob_h_ptr;   // [n_problems + 1] offsets into Hbuf / Abuf
  double* Hbuf; double* Abuf;
  pxr_lm_summary* summaries;   // device [n_problems]
  double det_scale;            // 0: floating-point atomics; 2^k: deterministic fixed-point accumulation of H and g (pxr_device.h)
  double* prob_scale;          // deterministic mode, [n_problems]: every sub-problem's grid (det_scale until its overflow guard asks for a coarser one)
  uint8_t* prob_done;          // deterministic mode, [n_problems]: the sub-problem finished (a repeated launch skips it)
  int* slot_of_node;           // [n_nodes] position of the node inside its sub-problem (the index of its LDS metadata)
  int stream_slots;            // > 0: the line-search probes run as ka_probe_stream (every sub-problem has at most this many nodes, <= 64,
                               // all its residual blocks in the LDS cache, no unary terms); its LDS region follows the LM state
  int stream_off;              // doubles from sh_A to that region
  double* prob_state = nullptr;     // [n_problems][KA_STATE]: LM state parked by a sub-problem that stopped with KA_TERM_RESCALE after an accepted step, or after its first iteration (KA_TERM_PARKED)
  int* sched = nullptr;             // two-phase launch (ka_solve_kernel_sched): [0] arrivals at the grid barrier, [1] release flag, [2] entries of the list, [3] next entry to take
  int* sched_list = nullptr;        // [n_problems] the parked sub-problems, those with a keypoint on a bound first
  uint8_t* prob_heavy = nullptr;    // [n_problems] set when a sub-problem is parked with a keypoint ON its bound
  int phase = 0;                    // the plain kernels (ka_solve_kernel_occ2): 0 the whole solve, 1 park after `park_iters` LM iterations, 2 resume the parked ones
  int park_iters = 1;
  const int* order = nullptr;       // phase 2: sub-problem of workgroup b (-1: none), the ones parked on a bound first
  // ---- label groups that span several workgroups (round 6) -----------------------------------------------------------------
  // A label group of the caller (ONE ceres::Problem of the reference: one trust region, one line search, one termination) may
  // be handed over as several CHUNKS -- consecutive sub-problems that share no variable (whole tracks each): pxr_ka_view.
  // d_prob_group.  Each chunk keeps its workgroup, its LDS state and its block-diagonal normal matrix; only the scalars a
  // Ceres problem decides on are summed over the group's workgroups (ka_group_sum4): cost, model cost change, the directional
  // derivative, every line-search probe's cost, step and parameter norms, the feasibility / rescale / factorisation flags.
  const int* grp_first = nullptr;   // [n_problems] first sub-problem of this sub-problem's group (NULL: every sub-problem is its own group)
  const int* grp_size = nullptr;    // [n_problems] sub-problems in the group
  double* grp_part = nullptr;       // [2][n_problems][4] the members' addends of the current / the previous sum
  unsigned* grp_cnt = nullptr;      // [n_problems][32] arrival counter of a group at its first member (one 128-byte line each)
};
constexpr int KA_GRP_CNT_STRIDE = 32;

// Sum of four block-uniform values over the workgroups of a label group; every member gets the SAME bits (partials are added in
// member order: lane l of the first wavefront takes members l, l + 64, ..., then a fixed shuffle tree).  `gen` counts the sums of
// this launch (the same in every member); two slots per member (gen & 1): a member can be at most one sum ahead of the slowest.
// All members are resident at the same time -- the host checks the group size against the launch's resident capacity, and
// sub-problems are dispatched in index order (the argument of k_chol_backsolve), so the spin-wait cannot deadlock.
struct KaGroup { int first, size, member, stride; double* part; unsigned* cnt; };
__device__ __noinline__ void ka_group_sum4(const KaGroup g, unsigned gen, double v0, double v1, double v2, double v3, double* sh_out) {
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double* slot = g.part + ((size_t)(gen & 1u) * g.stride + g.first) * 4;
    if (lane == 0) {
      double* mine = slot + (size_t)g.member * 4;
      mine[0] = v0; mine[1] = v1; mine[2] = v2; mine[3] = v3;
      __threadfence();
      unsigned* c = g.cnt + (size_t)g.first * KA_GRP_CNT_STRIDE;
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (gen + 1u) * (unsigned)g.size;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __builtin_amdgcn_wave_barrier();
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int m = lane; m < g.size; m += 64) {
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += __hip_atomic_load(slot + (size_t)m * 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
      if (lane == 0) sh_out[k] = s[k];
    }
  }
  __syncthreads();
}
constexpr int KA_STATE = 12;           // has_state, radius, iterations, accepted steps, initial cost, stencils, decrease factor, invalid steps,
                                       // cost, next fixed-point grid, reuse_diag, FULL (the linearisation itself was parked too)
constexpr int KA_TERM_RESCALE = 100;
constexpr int KA_TERM_PARKED = 101;    // internal: stopped after its first LM iteration (phase 1 of ka_solve_kernel_sched), resumes in phase 2   // internal termination code: the sub-problem's fixed-point grid did not fit, launch again

// Channel layout of a node over lanes: 8 channels per lane (one 16-byte fp16 load) for the CNN feature sizes, and the
// whole descriptor in ONE lane for CHANNELS < 8 -- the reference instantiates (128, 1) and (1, 1)
// (featuremetric_keypoint_optimizer.h:13-17) and takes the scalar all-fp64 [upstream] Ceres bicubic below 8 channels
// (interpolation.h:222-268), which interp_small restates.
template <int C> struct KaLay {
  static constexpr int CPL = C >= 8 ? 8 : C;     // channels per lane
  static constexpr int LPO = C / CPL;            // lanes per node / edge
};

// per-problem metadata the solve kernel keeps in LDS for its whole LM loop: every evaluation walks the same nodes and
// residual blocks, and the chains node -> patch -> scale / corner and slot -> edge -> endpoints -> unknowns were two to
// three dependent global loads in front of every interpolation / block (the kernel is latency-bound: one workgroup per
// sub-problem, ~18 % vector-ALU utilisation)
constexpr int KA_NODE_CACHE = 96, KA_EDGE_CACHE = 512;
struct KaNodeMeta { int64_t node, pi; double sx, sy, cx, cy; int64_t id; int v; };     // node < 0: not touched by a residual block; id: the node, v: its first unknown or -1
struct KaEdgeMeta { int n1, n2, v1, v2; double w; };

template <typename ST, int C, bool WITH_JAC>
__device__ __forceinline__ void ka_eval_node_at(const KaArgs& a, int64_t node, int64_t pi, double sx, double sy, double cx,
                                                double cy, const double* kp, int sub, bool fsimd);

template <typename ST, int C, bool WITH_JAC>
__device__ __forceinline__ void ka_eval_node(const KaArgs& a, int64_t node, const double* kp, int sub, bool fsimd) {
  const int64_t pi = a.v.d_node_patch[node];
  ka_eval_node_at<ST, C, WITH_JAC>(a, node, pi, a.scales[2 * pi], a.scales[2 * pi + 1], (double)a.corners[2 * pi],
                                   (double)a.corners[2 * pi + 1], kp, sub, fsimd);
}

template <typename ST, int C, bool WITH_JAC>
__device__ __forceinline__ void ka_eval_node_at(const KaArgs& a, int64_t node, int64_t pi, double sx, double sy, double cx,
                                                double cy, const double* kp, int sub, bool fsimd) {
  // FeaturePatch::ToPixelCoordinates, features/src/featurepatch.h:250-255
  const double u = kp[2 * node] * sx - 0.5 - cx;
  const double v = kp[2 * node + 1] * sy - 0.5 - cy;
  const ST* patch = reinterpret_cast<const ST*>(a.arena) + (size_t)pi * a.H * a.W * C;
  constexpr int CPL = KaLay<C>::CPL;
  double f[CPL], fr[CPL], fc[CPL];
  if constexpr (C >= 8) {
    if (fsimd) interp8<ST, C / 8, WITH_JAC, true>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f, fr, fc);
    else interp8<ST, C / 8, WITH_JAC, false>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f, fr, fc);
  } else {
    interp_small<ST, C>(patch, a.H, a.W, u, v, a.l2_normalize != 0, f, fr, fc);
  }
  // InterpolationConfig.check_bounds has NO effect on keypoint adjustment: PatchInterpolator::Evaluate returns the bounds
  // check (patch_interpolator.h:125-135), but FeatureMetric2DCostFunctor and FeatureReference2DCostFunctor ignore it and
  // return true (featuremetric.h:44-63, feature_reference.h:44-60); the box bounds of ParameterizeKeypoints keep the
  // keypoints inside their patches instead.
  double* d = a.desc + (size_t)node * 3 * C + sub * CPL;
#pragma unroll
  for (int ch = 0; ch < CPL; ++ch) {
    d[ch] = f[ch];
    if (WITH_JAC) { d[C + ch] = fc[ch] * sx; d[2 * C + ch] = fr[ch] * sy; }
  }
}

// a value every lane holds alike, moved to scalar registers (it stays live across the register-capped interpolation core)
__device__ __forceinline__ double uniform_f64(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]); u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

__device__ __forceinline__ double lpo_sum(double v, int LPO) { return LPO == 16 ? row16_sum(v) : (LPO == 8 ? row8_sum(v) : v); }

// block-wide sum, result broadcast to every thread (KA_NT threads)
__device__ __forceinline__ double block_sum(double v, double* sh4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = sh4[0];
#pragma unroll
  for (int k = 1; k < KA_NT / 64; ++k) t += sh4[k];
  return uniform_f64(t);     // every lane holds the sum: in scalar registers it does not compete with the interpolation core's 256
}

// block-wide exclusive prefix sum over the KA_NT threads (in thread order); total broadcast
__device__ __forceinline__ int block_excl_scan(int v, int* sh_scan, int& total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = v;
  for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
  __syncthreads();
  if (lane == 63) sh_scan[w] = x;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < w; ++k) base += sh_scan[k];
  total = 0;
#pragma unroll
  for (int k = 0; k < KA_NT / 64; ++k) total += sh_scan[k];
  return base + x - v;
}

// writes of one lane become visible to the other lanes of the same wavefront
__device__ __forceinline__ void wave_sync() {
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}

struct KaProb {
  int64_t np0, np1, ne0, ne1, nu0, nu1;
  int n;                 // unknowns
  int ncomp;             // connected components of the variable nodes
  double *g, *gun, *scale, *diag, *step, *delta, *lo, *hi, *rhs;
  int *row_off, *row_v0, *row_nc, *comp_v0;
  double* Hm;            // block-diagonal normal matrix: one dense nc x nc block per component
  const KaNodeMeta* cnode = nullptr;   // LDS: metadata of the first KA_NODE_CACHE nodes / KA_EDGE_CACHE edges of the problem
  const KaEdgeMeta* cedge = nullptr;   // (solve kernel only; the rest, and every other caller, reads global memory)
  double* csq = nullptr;               // LDS: squared residual norms of the cached edges (cost-only pass)
  double* ckp0 = nullptr;              // LDS: the current (accepted) keypoints of the cached nodes
  double* ckp = nullptr;               // LDS: the candidate keypoints of the cached nodes (ka_plus writes them here too: the probes'
                                       // node pass then starts without an L2 round trip in front of its texel addresses)
};
// DET: the kernel instantiation of the deterministic mode adds fixed-point integers only; the other floating-point atomics only
template <bool DET>
__device__ __forceinline__ void ka_accum(double* slot, double v, double det_scale) {
  if constexpr (DET) atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double2ll_rn(v * det_scale));
  else atomicAdd(slot, v);
}
// evaluate all nodes of the problem at keypoints `kp`
template <typename ST, int C, bool WITH_JAC>
__device__ void ka_nodes(const KaArgs& a, const KaProb& p, const double* kp, bool fsimd, bool moving_only = false) {
  constexpr int LPO = KaLay<C>::LPO, G = KA_NT / LPO;
  const int grp = threadIdx.x / LPO, sub = threadIdx.x % LPO;
  if constexpr (!WITH_JAC && C == 128 && sizeof(ST) <= 4) {
    // value-only pass over the cached nodes (the line-search probes): EIGHT lanes per node, two 8-channel chunks each -- 32
    // nodes per trip instead of 16 -- and, for a probe (moving_only), only the nodes that MOVE: a constant keypoint (a track's
    // root) keeps the descriptor the linearisation wrote
    if (p.cnode) {
      const int qgrp = threadIdx.x / 8, qsub = threadIdx.x % 8;
      const int64_t ncached = min((int64_t)KA_NODE_CACHE, p.np1 - p.np0);
#pragma nounroll
      for (int64_t i = p.np0 + qgrp; i - p.np0 < ncached; i += KA_NT / 8) {
        const KaNodeMeta m = p.cnode[i - p.np0];
        if (m.node < 0 || (moving_only && m.v < 0)) continue;
        const bool lds_kp = moving_only && p.ckp != nullptr;          // (a probe: kp is the candidate ka_plus just wrote)
        const double kx = lds_kp ? p.ckp[2 * (i - p.np0)] : kp[2 * m.node], ky = lds_kp ? p.ckp[2 * (i - p.np0) + 1] : kp[2 * m.node + 1];
        const double u = kx * m.sx - 0.5 - m.cx, v = ky * m.sy - 0.5 - m.cy;
        const ST* patch = reinterpret_cast<const ST*>(a.arena) + (size_t)m.pi * a.H * a.W * C;
        double fa[8], fb[8];
        if (fsimd) interp8x2_value<ST, true>(patch, a.H, a.W, qsub, u, v, a.l2_normalize != 0, fa, fb);
        else interp8x2_value<ST, false>(patch, a.H, a.W, qsub, u, v, a.l2_normalize != 0, fa, fb);
        double* d = a.desc + (size_t)m.node * 3 * C + qsub * 8;
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) { d[ch] = fa[ch]; d[64 + ch] = fb[ch]; }
      }
      for (int64_t i = p.np0 + KA_NODE_CACHE + grp; i < p.np1; i += G) {       // nodes beyond the cache: the general form
        const int64_t node = a.v.d_prob_nodes[i];
        if (a.used[node]) ka_eval_node<ST, C, WITH_JAC>(a, node, kp, sub, fsimd);
      }
      __syncthreads();
      return;
    }
  }
#pragma nounroll
  for (int64_t i = p.np0 + grp; i < p.np1; i += G) {
    if (p.cnode && i - p.np0 < KA_NODE_CACHE) {
      const KaNodeMeta m = p.cnode[i - p.np0];
      if (m.node < 0) continue;
      ka_eval_node_at<ST, C, WITH_JAC>(a, m.node, m.pi, m.sx, m.sy, m.cx, m.cy, kp, sub, fsimd);
      continue;
    }
    const int64_t node = a.v.d_prob_nodes[i];
    if (!a.used[node]) continue;
    ka_eval_node<ST, C, WITH_JAC>(a, node, kp, sub, fsimd);
  }
  __syncthreads();
}

// [upstream Ceres corrector.cc] kappa such that J~^T J~ = rho' (J^T J - kappa b b^T), b = J^T r
__device__ __forceinline__ double ka_kappa(double s, const double* rho) {
  if (s != 0.0 && rho[2] > 0.0) {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - sqrt(D);
    return (2.0 * alpha - alpha * alpha) / s;
  }
  return 0.0;
}

// walk the edges and the unary terms; returns the cost (block-uniform).  WITH_JAC: accumulates Hm
// and g (unscaled).
template <int C, bool WITH_JAC, bool DET = false>
__device__ double ka_terms(const KaArgs& a, const KaProb& p, double* sh4, const double det_scale = 0.0) {
  constexpr int LPO = KaLay<C>::LPO, CPL = KaLay<C>::CPL, G = KA_NT / LPO;
  const int grp = threadIdx.x / LPO, sub = threadIdx.x % LPO;
  double cost = 0.0;
  int64_t i_first = p.ne0 + grp;
  if constexpr (!WITH_JAC) {
    // cost only (the line-search probes: half of the solve's wall time).  The robust loss -- an fp64 log for Cauchy -- is
    // evaluated by ONE lane of a 16-lane group per residual block, yet costs the whole wavefront its ~150 instructions:
    // the groups only form the squared norms here and park them in LDS, then every lane evaluates the loss of a
    // different block.
    if (p.cedge && p.csq) {
      const int64_t ncached = min((int64_t)KA_EDGE_CACHE, p.ne1 - p.ne0);
      if constexpr (C >= 32) {
        // FOUR lanes per residual block here (C / 4 channels each, streamed): the pass is a chain of L2 round trips, not
        // arithmetic -- 64 blocks per trip instead of 16 cuts the trips of a 225-block sub-problem from 15 to 4
        constexpr int QL = 4, QG = KA_NT / QL, QC = C / QL;
        const int qgrp = threadIdx.x / QL, qsub = threadIdx.x % QL;
        for (int64_t i = p.ne0 + qgrp; i - p.ne0 < ncached; i += QG) {
          const KaEdgeMeta m = p.cedge[i - p.ne0];
          // lane q of the group takes channels 2 q + 8 k, 2 q + 8 k + 1: the four lanes read 64 consecutive bytes per load
          const double2* d1 = reinterpret_cast<const double2*>(a.desc + (size_t)m.n1 * 3 * C) + qsub;
          const double2* d2 = reinterpret_cast<const double2*>(a.desc + (size_t)m.n2 * 3 * C) + qsub;
          double2 x1[QC / 2], x2[QC / 2];
#pragma unroll
          for (int k = 0; k < QC / 2; ++k) { x1[k] = d1[QL * k]; x2[k] = d2[QL * k]; }
          double sq0 = 0, sq1 = 0;
#pragma unroll
          for (int k = 0; k < QC / 2; ++k) {
            const double r0 = x1[k].x - x2[k].x, r1 = x1[k].y - x2[k].y;
            sq0 = fma(r0, r0, sq0); sq1 = fma(r1, r1, sq1);
          }
          double sq = sq0 + sq1;
          sq += dpp_f64<0xB1>(sq);   // quad_perm [1,0,3,2]
          sq += dpp_f64<0x4E>(sq);   // quad_perm [2,3,0,1]
          if (qsub == 0) p.csq[i - p.ne0] = sq;
        }
        i_first = p.ne0 + ncached + grp;
      } else {
        for (; i_first - p.ne0 < ncached; i_first += G) {
          const KaEdgeMeta m = p.cedge[i_first - p.ne0];
          const double* d1 = a.desc + (size_t)m.n1 * 3 * C + sub * CPL;
          const double* d2 = a.desc + (size_t)m.n2 * 3 * C + sub * CPL;
          double sq = 0;
#pragma unroll
          for (int ch = 0; ch < CPL; ++ch) { const double r = d1[ch] - d2[ch]; sq = fma(r, r, sq); }
          sq = lpo_sum(sq, LPO);
          if (sub == 0) p.csq[i_first - p.ne0] = sq;
        }
      }
      __syncthreads();
      for (int64_t k = threadIdx.x; k < ncached; k += blockDim.x) {
        double rho[3];
        loss_eval(a.loss.type, a.loss.a, p.cedge[k].w, p.csq[k], rho);
        cost += 0.5 * rho[0];
      }
    }
  }
  if constexpr (WITH_JAC && C >= 64) {
    // linearisation, cached residual blocks: EIGHT lanes per block (C / 8 channels each as interleaved pairs: the eight lanes
    // read 128 consecutive bytes per load), 32 blocks per trip instead of 16 -- the trips are chains of L2 round trips
    if (p.cedge) {
      constexpr int QL = 8, QG = KA_NT / QL, QP = C / (2 * QL);       // QP channel pairs per lane
      const int qgrp = threadIdx.x / QL, qsub = threadIdx.x % QL;
      const int64_t ncached = min((int64_t)KA_EDGE_CACHE, p.ne1 - p.ne0);
#pragma nounroll
      for (int64_t i = p.ne0 + qgrp; i - p.ne0 < ncached; i += QG) {
        const KaEdgeMeta m = p.cedge[i - p.ne0];
        const double2* d1 = reinterpret_cast<const double2*>(a.desc + (size_t)m.n1 * 3 * C) + qsub;
        const double2* d2 = reinterpret_cast<const double2*>(a.desc + (size_t)m.n2 * 3 * C) + qsub;
        double q[14], s = 0;
#pragma unroll
        for (int k = 0; k < 14; ++k) q[k] = 0.0;
#pragma unroll
        for (int k = 0; k < QP; ++k) {
          const double2 f1 = d1[QL * k], f2 = d2[QL * k];
          const double2 gx1 = d1[C / 2 + QL * k], gy1 = d1[C + QL * k], gx2 = d2[C / 2 + QL * k], gy2 = d2[C + QL * k];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const double r = h ? f1.y - f2.y : f1.x - f2.x;
            const double a0 = h ? gx1.y : gx1.x, a1 = h ? gy1.y : gy1.x, a2 = -(h ? gx2.y : gx2.x), a3 = -(h ? gy2.y : gy2.x);
            s = fma(r, r, s);
            q[0] = fma(a0, a0, q[0]); q[1] = fma(a0, a1, q[1]); q[2] = fma(a0, a2, q[2]); q[3] = fma(a0, a3, q[3]);
            q[4] = fma(a1, a1, q[4]); q[5] = fma(a1, a2, q[5]); q[6] = fma(a1, a3, q[6]);
            q[7] = fma(a2, a2, q[7]); q[8] = fma(a2, a3, q[8]); q[9] = fma(a3, a3, q[9]);
            q[10] = fma(a0, r, q[10]); q[11] = fma(a1, r, q[11]); q[12] = fma(a2, r, q[12]); q[13] = fma(a3, r, q[13]);
          }
        }
        s = row8_sum(s);
#pragma unroll
        for (int k = 0; k < 14; ++k) q[k] = row8_sum(q[k]);
        double rho[3];
        loss_eval(a.loss.type, a.loss.a, m.w, s, rho);
        if (qsub == 0) {
          cost += 0.5 * rho[0];
          const double kappa = ka_kappa(s, rho);
          const int v1 = m.v1, v2 = m.v2;
          const int vv = v1 >= 0 ? v1 : v2;
          if (vv >= 0) {
            const int nc = p.row_nc[vv], c0 = p.row_v0[vv];
            const int idx[4] = {v1, v1 + 1, v2, v2 + 1};
            const bool var[4] = {v1 >= 0, v1 >= 0, v2 >= 0, v2 >= 0};
            const double b[4] = {q[10], q[11], q[12], q[13]};
            const double mm[4][4] = {{q[0], q[1], q[2], q[3]}, {q[1], q[4], q[5], q[6]}, {q[2], q[5], q[7], q[8]}, {q[3], q[6], q[8], q[9]}};
            double* blk = p.Hm + (p.row_off[vv] - (vv - c0) * nc);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
              if (!var[x]) continue;
              ka_accum<DET>(p.g + idx[x], rho[1] * b[x], det_scale);
#pragma unroll
              for (int y = 0; y < 4; ++y)
                if (var[y]) ka_accum<DET>(blk + (size_t)(idx[x] - c0) * nc + (idx[y] - c0), rho[1] * (mm[x][y] - kappa * b[x] * b[y]), det_scale);
            }
          }
        }
      }
      i_first = p.ne0 + ncached + grp;
    }
  }
#pragma nounroll
  for (int64_t i = i_first; i < p.ne1; i += G) {
    int n1, n2, v1c = 0, v2c = 0;
    double we;
    const bool cached = p.cedge && i - p.ne0 < KA_EDGE_CACHE;
    if (cached) {
      const KaEdgeMeta m = p.cedge[i - p.ne0];
      n1 = m.n1; n2 = m.n2; v1c = m.v1; v2c = m.v2; we = m.w;
    } else {
      const int e = a.v.d_prob_edges[i];
      n1 = a.v.d_edge_src[e]; n2 = a.v.d_edge_dst[e]; we = a.v.d_edge_w[e];
    }
    const double* d1 = a.desc + (size_t)n1 * 3 * C + sub * CPL;
    const double* d2 = a.desc + (size_t)n2 * 3 * C + sub * CPL;
    double r[CPL];
    double s = 0;
#pragma unroll
    for (int ch = 0; ch < CPL; ++ch) { r[ch] = d1[ch] - d2[ch]; s = fma(r[ch], r[ch], s); }
    s = lpo_sum(s, LPO);
    double rho[3];
    loss_eval(a.loss.type, a.loss.a, we, s, rho);
    if (sub == 0) cost += 0.5 * rho[0];
    if (WITH_JAC) {
      // J = [g1x g1y -g2x -g2y]; 10 entries of J^T J and 4 of J^T r
      double q[14];
#pragma unroll
      for (int k = 0; k < 14; ++k) q[k] = 0.0;
#pragma unroll
      for (int ch = 0; ch < CPL; ++ch) {
        const double a0 = d1[C + ch], a1 = d1[2 * C + ch], a2 = -d2[C + ch], a3 = -d2[2 * C + ch];
        q[0] = fma(a0, a0, q[0]); q[1] = fma(a0, a1, q[1]); q[2] = fma(a0, a2, q[2]); q[3] = fma(a0, a3, q[3]);
        q[4] = fma(a1, a1, q[4]); q[5] = fma(a1, a2, q[5]); q[6] = fma(a1, a3, q[6]);
        q[7] = fma(a2, a2, q[7]); q[8] = fma(a2, a3, q[8]); q[9] = fma(a3, a3, q[9]);
        q[10] = fma(a0, r[ch], q[10]); q[11] = fma(a1, r[ch], q[11]);
        q[12] = fma(a2, r[ch], q[12]); q[13] = fma(a3, r[ch], q[13]);
      }
#pragma unroll
      for (int k = 0; k < 14; ++k) q[k] = lpo_sum(q[k], LPO);
      if (sub == 0) {
        const double kappa = ka_kappa(s, rho);
        const int v1 = cached ? v1c : a.var_of_node[n1], v2 = cached ? v2c : a.var_of_node[n2];
        // both variable endpoints of an edge lie in the same component: one block, one column origin
        const int vv = v1 >= 0 ? v1 : v2;
        if (vv >= 0) {
          const int nc = p.row_nc[vv], c0 = p.row_v0[vv];
          const int idx[4] = {v1, v1 + 1, v2, v2 + 1};
          const bool var[4] = {v1 >= 0, v1 >= 0, v2 >= 0, v2 >= 0};
          const double b[4] = {q[10], q[11], q[12], q[13]};
          const double m[4][4] = {{q[0], q[1], q[2], q[3]}, {q[1], q[4], q[5], q[6]}, {q[2], q[5], q[7], q[8]}, {q[3], q[6], q[8], q[9]}};
          double* blk = p.Hm + (p.row_off[vv] - (vv - c0) * nc);
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            if (!var[x]) continue;
            ka_accum<DET>(p.g + idx[x], rho[1] * b[x], det_scale);
#pragma unroll
            for (int y = 0; y < 4; ++y)
              if (var[y]) ka_accum<DET>(blk + (size_t)(idx[x] - c0) * nc + (idx[y] - c0), rho[1] * (m[x][y] - kappa * b[x] * b[y]), det_scale);
          }
        }
      }
    }
  }
  // FeatureReference2DCostFunctor blocks: r = f - ref, J = [gx gy]
#pragma nounroll
  for (int64_t i = p.nu0 + grp; i < p.nu1; i += G) {
    const int u = a.v.d_prob_unary[i];
    const int n1 = a.v.d_unary_node[u];
    const double* d1 = a.desc + (size_t)n1 * 3 * C + sub * CPL;
    const double* rf = a.v.d_unary_ref + (size_t)u * C + sub * CPL;
    double r[CPL];
    double s = 0;
#pragma unroll
    for (int ch = 0; ch < CPL; ++ch) { r[ch] = d1[ch] - rf[ch]; s = fma(r[ch], r[ch], s); }
    s = lpo_sum(s, LPO);
    double rho[3];
    loss_eval(a.loss.type, a.loss.a, a.v.d_unary_w ? a.v.d_unary_w[u] : 1.0, s, rho);
    if (sub == 0) cost += 0.5 * rho[0];
    if (WITH_JAC) {
      double q[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int ch = 0; ch < CPL; ++ch) {
        const double a0 = d1[C + ch], a1 = d1[2 * C + ch];
        q[0] = fma(a0, a0, q[0]); q[1] = fma(a0, a1, q[1]); q[2] = fma(a1, a1, q[2]);
        q[3] = fma(a0, r[ch], q[3]); q[4] = fma(a1, r[ch], q[4]);
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) q[k] = lpo_sum(q[k], LPO);
      const int v1 = a.var_of_node[n1];
      if (sub == 0 && v1 >= 0) {
        const double kappa = ka_kappa(s, rho);
        const int nc = p.row_nc[v1], c0 = p.row_v0[v1];
        double* row = p.Hm + p.row_off[v1] + (v1 - c0);
        ka_accum<DET>(p.g + v1, rho[1] * q[3], det_scale);
        ka_accum<DET>(p.g + v1 + 1, rho[1] * q[4], det_scale);
        ka_accum<DET>(row, rho[1] * (q[0] - kappa * q[3] * q[3]), det_scale);
        ka_accum<DET>(row + 1, rho[1] * (q[1] - kappa * q[3] * q[4]), det_scale);
        ka_accum<DET>(row + nc, rho[1] * (q[1] - kappa * q[3] * q[4]), det_scale);
        ka_accum<DET>(row + nc + 1, rho[1] * (q[2] - kappa * q[4] * q[4]), det_scale);
      }
    }
  }
  return block_sum(cost, sh4);
}

// [upstream Ceres polynomial.cc] minimiser of the interpolating quadratic / cubic on [lo, hi]
__device__ double ka_poly(const double* c, int deg, double x) {
  double v = 0;
  for (int i = deg; i >= 0; --i) v = v * x + c[i];
  return v;
}
__device__ double ka_interp_step(double f0, double g0, bool have_prev, double xp, double fp, double xc, double fc,
                                 double lo, double hi) {
  double c[4] = {f0, g0, 0, 0};
  int deg;
  if (!have_prev) { c[2] = (fc - f0 - g0 * xc) / (xc * xc); deg = 2; }
  else {
    const double rp = (fp - f0 - g0 * xp) / (xp * xp), rc = (fc - f0 - g0 * xc) / (xc * xc);
    c[3] = (rc - rp) / (xc - xp); c[2] = rc - c[3] * xc; deg = 3;
  }
  double best_x = lo, best = ka_poly(c, deg, lo);
  double v = ka_poly(c, deg, hi);
  if (v < best) { best = v; best_x = hi; }
  if (deg == 2) {
    if (c[2] != 0.0) { const double x = -c[1] / (2 * c[2]); if (x > lo && x < hi && ka_poly(c, 2, x) < best) best_x = x; }
  } else {
    const double A = 3 * c[3], B = 2 * c[2], Cc = c[1];
    if (A == 0.0) {
      if (B != 0.0) { const double x = -Cc / B; if (x > lo && x < hi && ka_poly(c, 3, x) < best) best_x = x; }
    } else {
      const double disc = B * B - 4 * A * Cc;
      if (disc >= 0) {
        const double sq = sqrt(disc), x1 = (-B + sq) / (2 * A), x2 = (-B - sq) / (2 * A);
        if (x1 > lo && x1 < hi) { v = ka_poly(c, 3, x1); if (v < best) { best = v; best_x = x1; } }
        if (x2 > lo && x2 < hi) { v = ka_poly(c, 3, x2); if (v < best) { best = v; best_x = x2; } }
      }
    }
  }
  return best_x;
}

// candidate = P(x + alpha * delta) for the problem's variable nodes (ParameterBlock::Plus [upstream])
__device__ void ka_plus(const KaArgs& a, const KaProb& p, double alpha) {
  for (int64_t i = p.np0 + threadIdx.x; i < p.np1; i += blockDim.x) {
    const bool cached = p.cnode && i - p.np0 < KA_NODE_CACHE;
    const int64_t node = cached ? p.cnode[i - p.np0].id : a.v.d_prob_nodes[i];
    const int v = cached ? p.cnode[i - p.np0].v : a.var_of_node[node];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double val = (cached && p.ckp0) ? p.ckp0[2 * (i - p.np0) + c] : a.v.d_kp[2 * node + c];
      if (v >= 0) val = fmin(fmax(val + alpha * p.delta[v + c], p.lo[v + c]), p.hi[v + c]);
      a.kp_cand[2 * node + c] = val;
      if (cached && p.ckp) p.ckp[2 * (i - p.np0) + c] = val;
    }
  }
  __syncthreads();
}

// EXPERIMENT, compiled in with -DPXR_KA_STREAM_PROBE only (tools/variant_build.sh; off in the shipped library).  Measured at
// configs[1], round 5 (profiles/r5_ka_stream_probe.txt): a probe of a long line search takes 11.5 us instead of 15.8 us and the
// slowest sub-problem 2.2 ms instead of 2.5-2.8 ms -- but the whole solve 5.9-6.2 ms instead of 5.4 ms: the solve kernel sits at
// its 256-register cap, the out-of-line call (or, inlined, the second interpolation core) raises its spilled registers from 43 to
// 195-261 and every other phase of every sub-problem pays for them.  Kept as the starting point for a kernel whose phases are
// separate functions; bit-for-bit the same accepted steps (5852) and final cost to 1e-15.
#ifdef PXR_KA_STREAM_PROBE
// ---- the line-search probe as ONE streamed pass (cost only) ----------------------------------------------------------------
// A probe of the projected Armijo search ([upstream] trust_region_minimizer.cc DoLineSearch) needs the cost at
// P(x + alpha delta) and nothing else.  The general form (ka_plus + ka_nodes + ka_terms) is a chain of round trips: candidate
// keypoints to global memory, two trips of node interpolations, descriptors to the L2-resident scratch, four trips of residual
// blocks reading them back, the loss.  A sub-problem whose search direction runs into an active bound fails all twenty probes
// of its line search ([upstream] max_num_line_search_step_size_iterations) in several LM iterations -- ~125 probes per solve
// against ~5 for the others (tools/_ka_iter_hist.py) -- and those sub-problems are the tail of the launch.  Here the probe
// stays on the CU:
//   * FOUR lanes per node, every node of the sub-problem at once (<= 64), the descriptor in C / 32 steps of 32 channels: lane
//     `sub` interpolates channels 32 k + 8 sub .. + 7 with the arithmetic of interp8 / interp8x2_value (the four lanes read 64
//     consecutive bytes of every texel); with fp16 storage the texels of step k + 1 are requested before step k is computed;
//   * a step's 32 channels of every node go to LDS (rows of 34 doubles: consecutive nodes start 16 bytes apart in the bank
//     row), every residual block -- one lane each -- adds the step's part of f_a . f_b; |f|^2 accumulates with the node;
//   * |f_a / |f_a| - f_b / |f_b||^2 = |f_a|^2 n_a^2 + |f_b|^2 n_b^2 - 2 (f_a . f_b) n_a n_b,  n = 1 / |f|  (a constant node's
//     descriptor is the normalised one the linearisation left in the scratch: n = 1).  Fixed-order sums: every run alike.  The
//     difference from the channel-wise form is the rounding of three fp64 dot products, ~1e-16 of |f|^2 = 1.
// Only PixelInterpolator's L2-normalised descriptors (interpolation.h:642-677) take this path; the candidate keypoints are
// written (ka_plus) once the search has settled.
constexpr int KA_SCH = 32, KA_SROW = KA_SCH + 2, KA_STREAM_MAX = KA_NT / 4;
struct KaStream {                  // LDS
  double* buf;                     // [slots][KA_SROW] the current step's channels of every node
  double* self;                    // [slots] |f|^2 n^2
  double* ninv;                    // [slots] n
  double* kp;                      // [slots][2] the sub-problem's current keypoints
  const unsigned char* eslot;      // [edges][2] node slots of the cached residual blocks' endpoints
};

// (LDS pointers carry their address space here: out of line the compiler cannot infer it and would issue FLAT loads -- a quarter
// of the ds_read rate -- for the 128 row reads per lane and probe)
#define KA_LDS __attribute__((address_space(3)))
#define KA_GLOBAL __attribute__((address_space(1)))
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load (vmcnt(0)), i.e. for
// the texels of the NEXT step that are meant to stay in flight across the exchange (measured: 23 us per probe with it, slower
// than the general form's 16)
__device__ __forceinline__ void ka_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef PXR_KA_STREAM_FORCEINLINE     // A/B knob (tools/variant_build.sh)
#define KA_STREAM_INLINE __forceinline__
#else
#define KA_STREAM_INLINE __attribute__((noinline))
#endif
// (everything by value: a reference to the kernel's KaArgs / KaProb would force both structures into scratch memory for the whole
// kernel -- 1.4 KB per lane, every later field access a scratch load)
struct KaStreamCall {
  const KA_GLOBAL void* arena; const KA_GLOBAL double* desc;
  int H, W, loss_type, slots_max, nslots, ne;
  double loss_a;
  KA_LDS double* buf;                       // the region of KaStream: buf | self | ninv | kp
  const KA_LDS unsigned char* eslot;
  const KA_LDS double *delta, *lo, *hi;     // (the LM state is in LDS whenever this path is on)
  const KA_LDS KaNodeMeta* cnode; const KA_LDS KaEdgeMeta* cedge;
  KA_LDS double* sh4;
};
template <typename ST, int C, bool FS>
__device__ KA_STREAM_INLINE double ka_probe_stream(const KaStreamCall sc, double alpha) {
  static_assert(C % KA_SCH == 0, "whole steps");
  constexpr int NSTEP = C / KA_SCH;
  typedef typename Texel8<ST>::work_t HT;
  typedef double ldouble2 __attribute__((ext_vector_type(2)));
  KA_LDS double* const buf = sc.buf;
  KA_LDS double* const self = buf + sc.slots_max * KA_SROW;
  KA_LDS double* const sninv = self + sc.slots_max;
  const KA_LDS double* const skp = sninv + sc.slots_max;
  const KA_LDS unsigned char* const eslot = sc.eslot;
  const KA_LDS double* const delta = sc.delta;
  const KA_LDS double* const lo = sc.lo;
  const KA_LDS double* const hi = sc.hi;
  const KA_LDS KaNodeMeta* const cnode = sc.cnode;
  const KA_LDS KaEdgeMeta* const cedge = sc.cedge;
  const int tid = threadIdx.x, sub = tid & 3;
  const int nslots = sc.nslots, ne = sc.ne;
  const int slot = min(tid >> 2, nslots - 1);    // (idle lanes shadow the last node: valid addresses, nothing stored)
  const int64_t m_node = cnode[slot].node, m_pi = cnode[slot].pi, m_id = cnode[slot].id;
  const double m_sx = cnode[slot].sx, m_sy = cnode[slot].sy, m_cx = cnode[slot].cx, m_cy = cnode[slot].cy;
  const int m_v = cnode[slot].v;
  const bool active = (tid >> 2) < nslots && m_node >= 0, moving = active && m_v >= 0;
  const int vv = max(m_v, 0);
  // ParameterBlock::Plus [upstream]: the projected candidate, the expression of ka_plus
  const double x0 = skp[2 * slot], y0 = skp[2 * slot + 1];
  const double xc = fmin(fmax(x0 + alpha * delta[vv], lo[vv]), hi[vv]);
  const double yc = fmin(fmax(y0 + alpha * delta[vv + 1], lo[vv + 1]), hi[vv + 1]);
  const double kx = m_v >= 0 ? xc : x0, ky = m_v >= 0 ? yc : y0;
  const double u = kx * m_sx - 0.5 - m_cx, v = ky * m_sy - 0.5 - m_cy;        // featurepatch.h:250-255
  const KA_GLOBAL ST* const gpatch = (const KA_GLOBAL ST*)sc.arena + (size_t)m_pi * sc.H * sc.W * C;
  const StencilIndex si = stencil_index(sc.H, sc.W, u, v);
  // (global address space spelled out as well: a FLAT load counts on lgkmcnt too, so every LDS wait of the exchange would wait
  // for the texels of the next step -- 23 us per probe instead of 9)
  typedef unsigned int gu4 __attribute__((ext_vector_type(4)));
  typedef float gf4 __attribute__((ext_vector_type(4)));
  const KA_GLOBAL ldouble2* const cdesc = (const KA_GLOBAL ldouble2*)(sc.desc + (size_t)m_id * 3 * C + sub * 8);
  Texel8<ST> tx[4][4];
  ldouble2 cf[4];
  // every lane requests both -- its texels and the scratch row of its node -- and selects afterwards (a conditional load is a
  // branch and a wait per element)
  auto request = [&](int k) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const KA_GLOBAL ST* q = gpatch + (size_t)(si.ro[j] + si.co[i]) * C + k * KA_SCH + sub * 8;
        if constexpr (sizeof(ST) == 2) {
          const gu4 w = *(const KA_GLOBAL gu4*)q;
          tx[j][i].raw = make_uint4(w.x, w.y, w.z, w.w);
        } else {
          const gf4 w0 = *(const KA_GLOBAL gf4*)q, w1 = *(const KA_GLOBAL gf4*)(q + 4);
          tx[j][i].a = make_float4(w0.x, w0.y, w0.z, w0.w); tx[j][i].b = make_float4(w1.x, w1.y, w1.z, w1.w);
        }
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) cf[q] = cdesc[k * (KA_SCH / 2) + q];
  };
#define KA_STREAM_REQUEST(K) request(K)
  KA_STREAM_REQUEST(0);
  // this lane's residual blocks: tid and tid + KA_NT (at most KA_EDGE_CACHE = 2 KA_NT of them)
  const int e0 = tid, e1 = tid + KA_NT;
  const int sa0 = e0 < ne ? eslot[2 * e0] : 0, sb0 = e0 < ne ? eslot[2 * e0 + 1] : 0;
  const int sa1 = e1 < ne ? eslot[2 * e1] : 0, sb1 = e1 < ne ? eslot[2 * e1 + 1] : 0;
  double nrm2 = 0.0, dot0 = 0.0, dot1 = 0.0;
#pragma unroll
  for (int k = 0; k < NSTEP; ++k) {
    HT h[4][8], hd[4][8];
    double f[8], fr[8], fc[8];
    interp8_horizontal<ST, false, FS>(tx, si.dx, h, hd);
    ldouble2 cfk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cfk[q] = cf[q];
    // step k + 1 is requested as soon as the horizontal pass has consumed step k's texels: in flight behind the vertical pass,
    // the LDS exchange and its barriers (two steps' texels in registers cost the kernel 130 more spilled registers)
    if (k + 1 < NSTEP) KA_STREAM_REQUEST(k + 1);
    interp8_vertical<HT, false, FS>(h, hd, si.dy, f, fr, fc);
    if (!moving) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { f[2 * q] = cfk[q].x; f[2 * q + 1] = cfk[q].y; }
    }
    double s8 = 0.0;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) s8 = fma(f[ch], f[ch], s8);
    s8 += dpp_f64<0xB1>(s8);   // quad_perm [1,0,3,2]
    s8 += dpp_f64<0x4E>(s8);   // quad_perm [2,3,0,1]
    nrm2 += s8;
    if (active) {
      KA_LDS ldouble2* row = (KA_LDS ldouble2*)(buf + slot * KA_SROW + sub * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) { ldouble2 w2; w2.x = f[2 * q]; w2.y = f[2 * q + 1]; row[q] = w2; }
    }
    ka_lds_barrier();
    if (e0 < ne) {
      const KA_LDS ldouble2* ra = (const KA_LDS ldouble2*)(buf + sa0 * KA_SROW);
      const KA_LDS ldouble2* rb = (const KA_LDS ldouble2*)(buf + sb0 * KA_SROW);
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int q = 0; q < KA_SCH / 2; ++q) { const ldouble2 x = ra[q], y = rb[q]; d0 = fma(x.x, y.x, d0); d1 = fma(x.y, y.y, d1); }
      dot0 += d0 + d1;
    }
    if (e1 < ne) {
      const KA_LDS ldouble2* ra = (const KA_LDS ldouble2*)(buf + sa1 * KA_SROW);
      const KA_LDS ldouble2* rb = (const KA_LDS ldouble2*)(buf + sb1 * KA_SROW);
      double d0 = 0.0, d1 = 0.0;
#pragma unroll
      for (int q = 0; q < KA_SCH / 2; ++q) { const ldouble2 x = ra[q], y = rb[q]; d0 = fma(x.x, y.x, d0); d1 = fma(x.y, y.y, d1); }
      dot1 += d0 + d1;
    }
    ka_lds_barrier();
  }
#undef KA_STREAM_REQUEST
  if (active && sub == 0) {
    const double ninv = moving ? 1.0 / sqrt(nrm2) : 1.0;
    sninv[slot] = ninv; self[slot] = nrm2 * ninv * ninv;
  }
  __syncthreads();
  double cost = 0.0, rho[3];
  if (e0 < ne) {
    const double sq = self[sa0] + self[sb0] - 2.0 * dot0 * sninv[sa0] * sninv[sb0];
    loss_eval(sc.loss_type, sc.loss_a, cedge[e0].w, sq < 0.0 ? 0.0 : sq, rho);      // (a NaN stays a NaN)
    cost += 0.5 * rho[0];
  }
  if (e1 < ne) {
    const double sq = self[sa1] + self[sb1] - 2.0 * dot1 * sninv[sa1] * sninv[sb1];
    loss_eval(sc.loss_type, sc.loss_a, cedge[e1].w, sq < 0.0 ? 0.0 : sq, rho);
    cost += 0.5 * rho[0];
  }
  return block_sum(cost, (double*)sc.sh4);
}

#endif   // PXR_KA_STREAM_PROBE

// ---- Cholesky solves of one component block (n x n row-major lower, in place; b -> solution) ----
// whole workgroup (any n).  Returns false on a non-positive pivot (block-uniform).
__device__ bool ka_chol_block(double* A, int n, double* b) {
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  bool ok = true;
  for (int j = 0; j < n; ++j) {
    const double d = A[(size_t)j * n + j];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    const double piv = d > 0.0 ? sqrt(d) : 1.0, inv = 1.0 / piv;
    __syncthreads();
    if (tid == 0) A[(size_t)j * n + j] = piv;
    for (int i = j + 1 + tid; i < n; i += KA_NT) A[(size_t)i * n + j] *= inv;
    __syncthreads();
    for (int i = j + 1 + ti; i < n; i += KA_NT / 16) {
      const double lij = A[(size_t)i * n + j];
      for (int c = j + 1 + tj; c <= i; c += 16) A[(size_t)i * n + c] -= lij * A[(size_t)c * n + j];
    }
    __syncthreads();
  }
  for (int j = 0; j < n; ++j) {   // forward: L y = b (column oriented)
    const double yj = b[j] / A[(size_t)j * n + j];
    __syncthreads();
    if (tid == 0) b[j] = yj;
    for (int i = j + 1 + tid; i < n; i += KA_NT) b[i] -= A[(size_t)i * n + j] * yj;
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {   // backward: L^T x = y
    const double xj = b[j] / A[(size_t)j * n + j];
    __syncthreads();
    if (tid == 0) b[j] = xj;
    for (int i = tid; i < j; i += KA_NT) b[i] -= A[(size_t)j * n + i] * xj;
    __syncthreads();
  }
  return ok;
}

// one wavefront, 2 < n <= 64: 8 x 8 lanes over the trailing update, the right-hand side in registers
__device__ bool ka_chol_wave(double* A, int n, double* b) {
  const int lane = threadIdx.x & 63, ti = lane >> 3, tj = lane & 7;
  bool ok = true;
  for (int j = 0; j < n; ++j) {
    const double d = A[j * n + j];
    if (!(d > 0.0) || !isfinite(d)) ok = false;
    const double piv = d > 0.0 ? sqrt(d) : 1.0, inv = 1.0 / piv;
    wave_sync();
    if (lane == 0) A[j * n + j] = piv;
    { const int i = j + 1 + lane; if (i < n) A[i * n + j] *= inv; }
    wave_sync();
    for (int i = j + 1 + ti; i < n; i += 8) {
      const double lij = A[i * n + j];
      for (int c = j + 1 + tj; c <= i; c += 8) A[i * n + c] -= lij * A[c * n + j];
    }
    wave_sync();
  }
  double bi = lane < n ? b[lane] : 0.0;
  for (int j = 0; j < n; ++j) {
    const double yj = __shfl(bi, j) / A[j * n + j];
    if (lane == j) bi = yj;
    else if (lane > j && lane < n) bi -= A[lane * n + j] * yj;
  }
  for (int j = n - 1; j >= 0; --j) {
    const double xj = __shfl(bi, j) / A[j * n + j];
    if (lane == j) bi = xj;
    else if (lane < j) bi -= A[j * n + lane] * xj;
  }
  if (lane < n) b[lane] = bi;
  return ok;
}

// one wavefront, 2 < n <= KA_NREG: the block in REGISTERS, lane = row.  Lane r holds row r of the lower triangle; pivot j
// is read from lane j with v_readlane, every lane scales its own entry of column j, and the multipliers of the trailing
// update (the scaled entries of the other rows) come in with v_readlane again -- no LDS read-modify-write chains, no
// division: a track of 10 keypoints (18 unknowns) factors in ~5 k cycles against ~38 k for the LDS form above, which was
// a quarter of an LM iteration of the sub-problem.  The solves run on the registers too; the solution goes to b.
constexpr int KA_NREG = 24;
__device__ __forceinline__ double ka_readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// The block size as a template parameter: the run-time form (every row / column step under an `if (c < n)`: 276 wave-uniform branches
// with their scalar compares between the readlanes) spent ~9 us on an 18 x 18 block -- 20 us of every LM iteration of a
// 5-track sub-problem, the wavefront that takes two components running them back to back.
template <int N>
__device__ __attribute__((noinline)) bool ka_chol_reg_n(const double* A, double* b) {
  const int lane = threadIdx.x & 63;
  double row[N], inv[N];
  // (clamped address + select: a conditional load is a branch and a wait per element)
  const int rl = min(lane, N - 1);
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const double v = A[rl * N + c];
    row[c] = (lane < N && c <= lane) ? v : 0.0;
  }
  bool ok = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double d = ka_readlane_f64(row[j], j);
    const bool good = d > 0.0 && isfinite(d);
    ok = ok && good;
    inv[j] = good ? rsqrt(d) : 1.0;                           // 1 / L[j][j]
    const double l = row[j] * inv[j];                         // lane r >= j: L[r][j]  (lane j: sqrt(d))
    row[j] = l;
#pragma unroll
    for (int c = j + 1; c < N; ++c) row[c] = fma(-l, ka_readlane_f64(l, c), row[c]);   // rows r >= c use it; the others hold dead entries
  }
  double bi = lane < N ? b[lane] : 0.0;
#pragma unroll
  for (int j = 0; j < N; ++j) {                               // L y = b
    const double yj = ka_readlane_f64(bi, j) * inv[j];
    bi = lane == j ? yj : (lane > j ? fma(-row[j], yj, bi) : bi);
  }
  // L^T x = y: step j needs L[j][r] in lane r -- lane j's row, one entry per lane: through the block's own storage (A is scratch
  // of the iteration: overwritten with the factor's rows; a lane also stores the dead entries right of its diagonal: nothing
  // reads the block's upper triangle), and the column of L^T a lane needs is loaded in one go, not entry by entry inside the
  // substitution
  double* Aw = const_cast<double*>(A);
  if (lane < N) {
#pragma unroll
    for (int c = 0; c < N; ++c) Aw[lane * N + c] = row[c];
  }
  wave_sync();
  double col[N];
#pragma unroll
  for (int j = 0; j < N; ++j) col[j] = Aw[j * N + rl];
#pragma unroll
  for (int j = N - 1; j >= 0; --j) {
    const double xj = ka_readlane_f64(bi, j) * inv[j];
    bi = lane == j ? xj : (lane < j ? fma(-col[j], xj, bi) : bi);
  }
  if (lane < N) b[lane] = bi;
  return ok;
}
// 2 < n <= KA_NREG, n even (two unknowns per keypoint)
__device__ __forceinline__ bool ka_chol_wave_reg(const double* A, int n, double* b) {
  switch (n) {
    case 4: return ka_chol_reg_n<4>(A, b);
    case 6: return ka_chol_reg_n<6>(A, b);
    case 8: return ka_chol_reg_n<8>(A, b);
    case 10: return ka_chol_reg_n<10>(A, b);
    case 12: return ka_chol_reg_n<12>(A, b);
    case 14: return ka_chol_reg_n<14>(A, b);
    case 16: return ka_chol_reg_n<16>(A, b);
    case 18: return ka_chol_reg_n<18>(A, b);
    case 20: return ka_chol_reg_n<20>(A, b);
    case 22: return ka_chol_reg_n<22>(A, b);
    case 24: return ka_chol_reg_n<24>(A, b);
    default: return ka_chol_wave(const_cast<double*>(A), n, b);      // (cannot happen: the unknowns come in pairs)
  }
}

// one thread, n == 2 (a keypoint that is alone in its component)
__device__ __forceinline__ bool ka_chol_2x2(const double* A, double* b) {
  const double a00 = A[0], a10 = A[2], a11 = A[3];
  bool ok = a00 > 0.0 && isfinite(a00);
  const double l00 = a00 > 0.0 ? sqrt(a00) : 1.0, l10 = a10 / l00, d1 = a11 - l10 * l10;
  ok = ok && d1 > 0.0 && isfinite(d1);
  const double l11 = d1 > 0.0 ? sqrt(d1) : 1.0;
  const double y0 = b[0] / l00, y1 = (b[1] - l10 * y0) / l11;
  const double x1 = y1 / l11, x0 = (y0 - l10 * x1) / l00;
  b[0] = x0; b[1] = x1;
  return ok;
}

// ---- setup: one workgroup per sub-problem finds the connected components of its variable nodes,
// lays the unknowns out component by component and writes the box bounds.  Its per-problem block
// sizes let the host size the normal-matrix storage exactly before the solve kernel starts.
struct KaInfo { int n, ncomp, hsz, maxnc, feasible; };

__global__ __launch_bounds__(KA_NT) void ka_setup_kernel(const KaArgs a, KaInfo* __restrict__ info) {
  __shared__ int sh_scan[KA_NT / 64];
  __shared__ int sh_flag, sh_feasible, sh_maxnc;
  const int prob = blockIdx.x, tid = threadIdx.x;
  KaProb p;
  p.np0 = a.v.d_prob_node_ptr[prob]; p.np1 = a.v.d_prob_node_ptr[prob + 1];
  p.ne0 = a.v.d_prob_edge_ptr[prob]; p.ne1 = a.v.d_prob_edge_ptr[prob + 1];
  p.nu0 = p.nu1 = 0;
  if (a.v.n_unary > 0) { p.nu0 = a.v.d_prob_unary_ptr[prob]; p.nu1 = a.v.d_prob_unary_ptr[prob + 1]; }
  const size_t vstride = 2 * (size_t)a.v.n_nodes, vb = 2 * (size_t)p.np0;
  p.lo = a.vec + 6 * vstride + vb; p.hi = a.vec + 7 * vstride + vb;
  p.row_off = a.irow + 0 * vstride + vb; p.row_v0 = a.irow + 1 * vstride + vb; p.row_nc = a.irow + 2 * vstride + vb;
  p.comp_v0 = a.comp_v0 + p.np0;
  const int nloc = (int)(p.np1 - p.np0);
  int* cnt = a.ipos + 0 * (size_t)a.v.n_nodes + p.np0;
  int* cstart = a.ipos + 1 * (size_t)a.v.n_nodes + p.np0;
  int* hoff = a.ipos + 2 * (size_t)a.v.n_nodes + p.np0;
  int* cidx = a.ipos + 3 * (size_t)a.v.n_nodes + p.np0;
  // will_be_optimized_ (featuremetric_keypoint_optimizer.h:198-199): nodes touched by a residual block
  for (int64_t i = p.ne0 + tid; i < p.ne1; i += blockDim.x) {
    const int e = a.v.d_prob_edges[i];
    a.used[a.v.d_edge_src[e]] = 1; a.used[a.v.d_edge_dst[e]] = 1;
  }
  for (int64_t i = p.nu0 + tid; i < p.nu1; i += blockDim.x) a.used[a.v.d_unary_node[a.v.d_prob_unary[i]]] = 1;
  if (tid == 0) { sh_feasible = 1; sh_maxnc = 0; }
  __syncthreads();
  // ---- connected components of the variable nodes (min-label propagation over the edges) ----
  for (int i = tid; i < nloc; i += blockDim.x) {
    const int64_t node = a.v.d_prob_nodes[p.np0 + i];
    a.label[node] = (a.used[node] && a.v.d_node_const[node] != 1) ? i : -1;
#ifdef PXR_KA_STREAM_PROBE
    a.slot_of_node[node] = i;
#endif
    cnt[i] = 0;
  }
  __syncthreads();
  while (true) {
    if (tid == 0) sh_flag = 0;
    __syncthreads();
    for (int64_t i = p.ne0 + tid; i < p.ne1; i += blockDim.x) {
      const int e = a.v.d_prob_edges[i];
      const int n1 = a.v.d_edge_src[e], n2 = a.v.d_edge_dst[e];
      const int l1 = a.label[n1], l2 = a.label[n2];
      if (l1 >= 0 && l2 >= 0 && l1 != l2) {
        const int m = min(l1, l2);
        atomicMin(&a.label[n1], m); atomicMin(&a.label[n2], m);
        sh_flag = 1;
      }
    }
    __syncthreads();
    for (int i = tid; i < nloc; i += blockDim.x) {   // pointer jumping towards the component's first node
      const int64_t node = a.v.d_prob_nodes[p.np0 + i];
      const int l = a.label[node];
      if (l >= 0) {
        const int r = a.label[a.v.d_prob_nodes[p.np0 + l]];
        if (r < l) { atomicMin(&a.label[node], r); sh_flag = 1; }
      }
    }
    __syncthreads();
    const int again = sh_flag;
    __syncthreads();
    if (!again) break;
  }
  for (int i = tid; i < nloc; i += blockDim.x) {
    const int l = a.label[a.v.d_prob_nodes[p.np0 + i]];
    if (l >= 0) atomicAdd(&cnt[l], 1);
  }
  __syncthreads();
  // component order = order of first nodes; scans give each component's first unknown, block offset, index
  int run_nodes = 0, run_h = 0, run_c = 0;
  for (int base = 0; base < nloc; base += KA_NT) {
    const int i = base + tid;
    const int c = i < nloc ? cnt[i] : 0;
    int tot;
    const int e_nodes = block_excl_scan(c, sh_scan, tot) + run_nodes; run_nodes += tot;
    const int e_h = block_excl_scan(4 * c * c, sh_scan, tot) + run_h; run_h += tot;
    const int e_c = block_excl_scan(c > 0 ? 1 : 0, sh_scan, tot) + run_c; run_c += tot;
    if (c > 0) {
      cstart[i] = e_nodes; hoff[i] = e_h; cidx[i] = e_c;
      p.comp_v0[e_c] = 2 * e_nodes;
      atomicMax(&sh_maxnc, 2 * c);
    }
  }
  __syncthreads();
  const int n = 2 * run_nodes, hsz = run_h;
  // unknown layout: components in order, nodes inside a component in ascending order; box bounds
  // (keypoint_optimizer.h:127-152)
  for (int i = tid; i < nloc; i += blockDim.x) {
    const int64_t node = a.v.d_prob_nodes[p.np0 + i];
    const int l = a.label[node];
    int v = -1;
    if (l >= 0) {
      int rank = 0;
      for (int j = l; j < i; ++j) rank += (a.label[a.v.d_prob_nodes[p.np0 + j]] == l) ? 1 : 0;
      const int v0 = 2 * cstart[l], nc = 2 * cnt[l];
      v = v0 + 2 * rank;
      p.row_v0[v] = v0; p.row_v0[v + 1] = v0;
      p.row_nc[v] = nc; p.row_nc[v + 1] = nc;
      p.row_off[v] = hoff[l] + (v - v0) * nc; p.row_off[v + 1] = hoff[l] + (v + 1 - v0) * nc;
      const int64_t pi = a.v.d_node_patch[node];
      const double sx = a.scales[2 * pi], sy = a.scales[2 * pi + 1];
      const double kx = a.v.d_kp[2 * node], ky = a.v.d_kp[2 * node + 1];
      double lx = (a.corners[2 * pi] + 0.5) / sx, ly = (a.corners[2 * pi + 1] + 0.5) / sy;
      double ux = lx + a.W / sx, uy = ly + a.H / sy;
      if (a.bound > 0.0) {
        ux = fmin(kx + a.bound / sx, ux); uy = fmin(ky + a.bound / sy, uy);
        lx = fmax(kx - a.bound / sx, lx); ly = fmax(ky - a.bound / sy, ly);
      }
      if (a.v.d_node_const[node] == 2) {   // a destination keypoint outside nodes_in_problem: ParameterizeKeypoints
        lx = ly = -INFINITY; ux = uy = INFINITY;   // (keypoint_optimizer.h:117) never visits it -- free, no bounds
      }
      p.lo[v] = lx; p.lo[v + 1] = ly; p.hi[v] = ux; p.hi[v + 1] = uy;
      if (kx < lx || kx > ux || ky < ly || ky > uy) sh_feasible = 0;
    }
    a.var_of_node[node] = v;
  }
  __syncthreads();
  if (tid == 0) {
    KaInfo o;
    o.n = n; o.ncomp = run_c; o.hsz = hsz; o.maxnc = sh_maxnc; o.feasible = sh_feasible;
    info[prob] = o;
  }
}


// -DPXR_KA_PROFILE: workgroup 0 prints how its wall time splits over the phases of the LM loop (tools/ka_phase_probe.sh)
#ifdef PXR_KA_PROFILE
#ifndef PXR_KA_PROFILE_BLOCK
#define PXR_KA_PROFILE_BLOCK 0      // the sub-problem that reports (tools/ka_phase_probe.sh [block])
#endif
#define KA_T(k) do { if (blockIdx.x == PXR_KA_PROFILE_BLOCK && threadIdx.x == 0) { const long long t_ = wall_clock64(); ka_prof[k] += t_ - ka_t0; ka_t0 = t_; } } while (0)
#else
#define KA_T(k) do { } while (0)
#endif


// the deterministic linearisation's epilogue, OUT of line (it must not weigh on the register allocation of the interpolation core):
// slots -> doubles, the overflow guard, the next grid.  Returns the next grid; *rescale != 0: launch again with that grid.
__device__ __attribute__((noinline)) double ka_finish_fixed_point(double* Hm, double* g, const int* row_off, const int* row_v0, int hsz, int n,
                                                                  double grid, double c, bool first, double* sh4, double* rescale) {
  const int tid = threadIdx.x;
  double dsum = 0.0;
  const double inv_grid = 1.0 / grid;          // (a power of two: exact)
  for (int e = tid; e < hsz; e += blockDim.x) Hm[e] = (double)__double_as_longlong(Hm[e]) * inv_grid;
  for (int e = tid; e < n; e += blockDim.x) g[e] = (double)__double_as_longlong(g[e]) * inv_grid;
  __syncthreads();
  for (int e = tid; e < n; e += blockDim.x) {
    const double d = Hm[row_off[e] + e - row_v0[e]];
    dsum += d < 0.0 ? NAN : d;                                   // (a wrapped diagonal slot poisons the sum)
  }
  const double tr = block_sum(dsum, sh4);
  const double bound = fmax(tr, sqrt(2.0 * tr * fmax(c, 0.0)));
  const bool usable = isfinite(bound) && bound > 0x1p-900 && bound < 0x1p900;
  double ideal = grid;
  if (usable) {
    const long long bits = __double_as_longlong(bound);
    const int e2 = (int)((bits >> 52) & 0x7ff) - 1023 + ((bits & 0xfffffffffffffll) != 0 ? 1 : 0);       // ceil(log2(bound))
    ideal = __longlong_as_double((long long)(1023 + 58 - e2) << 52);                                     // 2^(58 - e2)
  }
  if (isnan(tr) && isfinite(c)) *rescale = grid * 0x1p-16;
  else if (usable && (bound * grid > 0x1p62 || (first && grid < ideal * 0x1p-10))) *rescale = ideal;
  return ideal;
}

// phase 0: the whole solve; 1: at most ONE LM iteration, then the LM state is parked (KA_TERM_PARKED); 2: resume a parked sub-problem
template <typename ST, int C, bool DET>
__device__ __forceinline__ void ka_solve_body(const KaArgs& a, const KaInfo* __restrict__ info, double* sh_A, const int prob, const int phase) {
#ifdef PXR_KA_PROFILE
  long long ka_prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ka_t0 = wall_clock64();
  const long long ka_tstart = ka_t0;
#endif
  __shared__ double sh4[KA_NT / 64];
  __shared__ int sh_ok;
  const int tid = threadIdx.x;
  const bool fsimd = a.float_simd != 0;
  KaProb p;
  if (DET && a.prob_done[prob]) return;                  // (a repeated launch after a grid change: this sub-problem had finished)
  // a label group that spans several workgroups (KaArgs::grp_first): the Ceres-level scalars are summed over its members
  __shared__ double sh_gs[4];
  KaGroup grp{prob, 1, 0, 0, nullptr, nullptr};
  if (a.grp_first) { grp.first = a.grp_first[prob]; grp.size = a.grp_size[prob]; grp.member = prob - grp.first; grp.stride = a.v.n_problems; grp.part = a.grp_part; grp.cnt = a.grp_cnt; }
  const bool grouped = grp.size > 1;
  unsigned grp_gen = 0;
  auto gsum4 = [&](double& v0, double& v1, double& v2, double& v3) {       // in place; a no-op for a group of one
    if (!grouped) return;
    ka_group_sum4(grp, grp_gen++, v0, v1, v2, v3, sh_gs);
    v0 = uniform_f64(sh_gs[0]); v1 = uniform_f64(sh_gs[1]); v2 = uniform_f64(sh_gs[2]); v3 = uniform_f64(sh_gs[3]);
  };
  __shared__ double sh_grid, sh_resc;
  if (tid == 0) { sh_grid = DET ? a.prob_scale[prob] : 0.0; sh_resc = 0.0; }

  p.np0 = a.v.d_prob_node_ptr[prob]; p.np1 = a.v.d_prob_node_ptr[prob + 1];
  p.ne0 = a.v.d_prob_edge_ptr[prob]; p.ne1 = a.v.d_prob_edge_ptr[prob + 1];
  p.nu0 = p.nu1 = 0;
  if (a.v.n_unary > 0) { p.nu0 = a.v.d_prob_unary_ptr[prob]; p.nu1 = a.v.d_prob_unary_ptr[prob + 1]; }
  const size_t vstride = 2 * (size_t)a.v.n_nodes, vb = 2 * (size_t)p.np0;
  p.g = a.vec + 0 * vstride + vb; p.gun = a.vec + 1 * vstride + vb; p.scale = a.vec + 2 * vstride + vb;
  p.diag = a.vec + 3 * vstride + vb; p.step = a.vec + 4 * vstride + vb; p.delta = a.vec + 5 * vstride + vb;
  p.lo = a.vec + 6 * vstride + vb; p.hi = a.vec + 7 * vstride + vb; p.rhs = a.vec + 8 * vstride + vb;
  p.row_off = a.irow + 0 * vstride + vb; p.row_v0 = a.irow + 1 * vstride + vb; p.row_nc = a.irow + 2 * vstride + vb;
  p.comp_v0 = a.comp_v0 + p.np0;
  p.Hm = a.Hbuf + a.prob_h_ptr[prob];
  pxr_lm_summary sm;
  sm.iterations = 0; sm.num_successful = 0; sm.termination = PXR_TERM_NO_CONVERGENCE;
  sm.num_camera_unknowns = 0; sm.num_point_unknowns = 0; sm.initial_cost = 0; sm.final_cost = 0;
  sm.final_radius = 0; sm.total_ms = 0; sm.setup_ms = 0; sm.linear_solver = 0; sm.collective_kib = 0; sm.linear_iterations = 0; sm.accumulation = DET ? 1 : 0; sm.initial_us = 0;

  __shared__ KaNodeMeta sh_nodes[KA_NODE_CACHE];
  __shared__ KaEdgeMeta sh_edges[KA_EDGE_CACHE];
  __shared__ double sh_sq[KA_EDGE_CACHE];
  __shared__ double sh_kpc[2 * KA_NODE_CACHE], sh_kp0[2 * KA_NODE_CACHE];
#ifdef PXR_KA_STREAM_PROBE
  __shared__ unsigned char sh_eslot[2 * KA_EDGE_CACHE];     // node slots of the cached residual blocks' endpoints (ka_probe_stream)
#endif
  for (int64_t i = p.np0 + tid; i < p.np1 && i - p.np0 < KA_NODE_CACHE; i += blockDim.x) {
    const int64_t node = a.v.d_prob_nodes[i];
    const int64_t pi = a.v.d_node_patch[node];
    KaNodeMeta m;
    m.node = a.used[node] ? node : -1; m.pi = pi; m.id = node; m.v = a.var_of_node[node];
    m.sx = a.scales[2 * pi]; m.sy = a.scales[2 * pi + 1]; m.cx = (double)a.corners[2 * pi]; m.cy = (double)a.corners[2 * pi + 1];
    sh_nodes[i - p.np0] = m;
    sh_kp0[2 * (i - p.np0)] = a.v.d_kp[2 * node]; sh_kp0[2 * (i - p.np0) + 1] = a.v.d_kp[2 * node + 1];
  }
  for (int64_t i = p.ne0 + tid; i < p.ne1 && i - p.ne0 < KA_EDGE_CACHE; i += blockDim.x) {
    const int e = a.v.d_prob_edges[i];
    KaEdgeMeta m;
    m.n1 = a.v.d_edge_src[e]; m.n2 = a.v.d_edge_dst[e]; m.v1 = a.var_of_node[m.n1]; m.v2 = a.var_of_node[m.n2]; m.w = a.v.d_edge_w[e];
    sh_edges[i - p.ne0] = m;
#ifdef PXR_KA_STREAM_PROBE
    sh_eslot[2 * (i - p.ne0)] = (unsigned char)a.slot_of_node[m.n1];       // (meaningful in sub-problems of <= 64 nodes: the
    sh_eslot[2 * (i - p.ne0) + 1] = (unsigned char)a.slot_of_node[m.n2];   // only ones that read it)
#endif
  }
  p.cnode = sh_nodes; p.cedge = sh_edges; p.csq = sh_sq; p.ckp = sh_kpc; p.ckp0 = sh_kp0;
  __syncthreads();

  const KaInfo inf = info[prob];
  const int n = inf.n, hsz = inf.hsz, maxnc = inf.maxnc;
  p.n = n; p.ncomp = inf.ncomp;
  sm.num_camera_unknowns = n;
  double* A = (hsz <= a.lds_elems) ? sh_A : (a.Abuf + a.prob_h_ptr[prob]);
  if (a.lds_state_n > 0) {
    // The LM state of the sub-problem in LDS for the whole solve: the normal matrix the residual blocks are accumulated
    // into (20 atomics per block went to L2 before) and the vectors / index arrays every short phase of an iteration
    // walks -- each phase paid an L2 round trip for a few hundred bytes.  Launch-time decision: every sub-problem fits.
    const int nm = a.lds_state_n;
    double* base = sh_A + a.lds_elems;
    p.Hm = base; base += a.lds_elems;
    double* const lo_g = p.lo; double* const hi_g = p.hi;
    const int *ro_g = p.row_off, *rv_g = p.row_v0, *rn_g = p.row_nc, *cv_g = p.comp_v0;
    p.g = base; p.gun = base + nm; p.scale = base + 2 * nm; p.diag = base + 3 * nm; p.step = base + 4 * nm;
    p.delta = base + 5 * nm; p.lo = base + 6 * nm; p.hi = base + 7 * nm;
    int* ib = reinterpret_cast<int*>(base + 8 * nm);
    p.row_off = ib; p.row_v0 = ib + nm; p.row_nc = ib + 2 * nm; p.comp_v0 = ib + 3 * nm;
    for (int e = tid; e < n; e += blockDim.x) {
      p.lo[e] = lo_g[e]; p.hi[e] = hi_g[e];
      p.row_off[e] = ro_g[e]; p.row_v0[e] = rv_g[e]; p.row_nc[e] = rn_g[e];
    }
    for (int c = tid; c < inf.ncomp; c += blockDim.x) p.comp_v0[c] = cv_g[c];
    __syncthreads();
  }
  const pxr_lm_options& opt = a.opt;
  // the line-search probes as one streamed pass (ka_probe_stream) when the launch was sized for it and this sub-problem fits
  bool use_stream = false;
#ifdef PXR_KA_STREAM_PROBE
  KaStream strm{};
  KaStreamCall scall{};
  if constexpr (C % KA_SCH == 0 && sizeof(ST) <= 4) {
    if (a.stream_slots > 0 && p.np1 - p.np0 <= a.stream_slots && p.ne1 - p.ne0 <= KA_EDGE_CACHE && p.nu1 == p.nu0) {
      use_stream = true;
      strm.buf = sh_A + a.stream_off; strm.self = strm.buf + (size_t)a.stream_slots * KA_SROW;
      strm.ninv = strm.self + a.stream_slots; strm.kp = strm.ninv + a.stream_slots; strm.eslot = sh_eslot;
      scall.arena = (const KA_GLOBAL void*)a.arena; scall.desc = (const KA_GLOBAL double*)a.desc;
      scall.H = a.H; scall.W = a.W; scall.loss_type = a.loss.type; scall.loss_a = a.loss.a; scall.slots_max = a.stream_slots;
      scall.nslots = (int)(p.np1 - p.np0); scall.ne = (int)(p.ne1 - p.ne0);
      scall.buf = (KA_LDS double*)strm.buf; scall.eslot = (const KA_LDS unsigned char*)sh_eslot;
      scall.delta = (const KA_LDS double*)p.delta; scall.lo = (const KA_LDS double*)p.lo; scall.hi = (const KA_LDS double*)p.hi;
      scall.cnode = (const KA_LDS KaNodeMeta*)sh_nodes; scall.cedge = (const KA_LDS KaEdgeMeta*)sh_edges; scall.sh4 = (KA_LDS double*)sh4;
      for (int i = tid; i < (int)(p.np1 - p.np0); i += blockDim.x) {
        const int64_t node = sh_nodes[i].id;
        strm.kp[2 * i] = a.v.d_kp[2 * node]; strm.kp[2 * i + 1] = a.v.d_kp[2 * node + 1];
      }
      __syncthreads();
    }
  }
#endif

  // evaluate cost + normal equations at the CURRENT keypoints, then scale: H <- S H S, g <- S g
  // node stencils interpolated over the solve (a linearisation evaluates every node, a line-search probe the variable ones):
  // the algorithmic traffic of the solve is 16 texels x C x sizeof(storage) each, reported as summary.linear_iterations
  int64_t stencils = 0;
  const int64_t nodes_all = p.np1 - p.np0, nodes_var = n / 2;
  double grid_used = 0.0;              // deterministic mode: the fixed-point grid the newest linearisation was accumulated on
  auto linearize = [&](const bool compute_scale) -> double {        // compute_scale: the FIRST linearisation of the solve
    if (DET) grid_used = uniform_f64(sh_grid);
    stencils += nodes_all;
    for (int e = tid; e < hsz; e += blockDim.x) p.Hm[e] = 0.0;
    for (int e = tid; e < n; e += blockDim.x) p.g[e] = 0.0;
    __syncthreads();
    KA_T(7);
    ka_nodes<ST, C, true>(a, p, a.v.d_kp, fsimd);
    KA_T(0);
    const double c = ka_terms<C, true, DET>(a, p, sh4, DET ? sh_grid : 0.0);
    __syncthreads();
    KA_T(1);
    if constexpr (DET) {
      // The fixed-point grid of the slots follows the data.  Integer atomics wrap modulo 2^64, so only the FINAL content of a
      // slot has to fit: |H_ab| <= max diag(H) <= trace(H) (each block's J^T J with the corrector is positive semi-definite),
      // |g_a| <= sqrt(H_aa 2 cost) (Cauchy-Schwarz; rho concave: rho' s <= rho), so bound = max(trace, sqrt(2 trace cost)).  The
      // trace is read off the finished diagonal: its addends are all >= 0, a diagonal slot that wrapped shows as a negative
      // entry (it would take 2^64 units -- 64x the bound of the grid AFTER the 16x headroom -- to come round to a plausible value).
      // The ideal grid puts the bound at 2^57 .. 2^58 units, a factor 16 below the 2^62 limit; every linearisation runs on
      // the ideal grid of the one before.  The sub-problem stops and asks the host for another launch (sh_resc = the grid
      // to use) if a diagonal entry is negative, if the bound grew by more than those 16x in one accepted step, or if the start
      // grid (2^-38: unit-norm 128-channel descriptors) does not suit the FIRST linearisation -- it overflows (raw features) or
      // is more than 2^10 coarser than ideal (single-channel features: H ~ 1e-4, g -> 1e-9 at convergence).  One workgroup,
      // a static block -> lane mapping, fixed reduction trees: the same decision on every run.  (The grid and the request live in
      // LDS and the epilogue is out of line: as loop-carried registers / inline code they cost the interpolation core 20 more
      // spilled registers; what the mode costs with them there: profiles/r5_ka_det_probes.txt.)
      double resc = 0.0;
      const double ideal = ka_finish_fixed_point(p.Hm, p.g, p.row_off, p.row_v0, hsz, n, sh_grid, c, compute_scale, sh4, &resc);
      if (tid == 0) { sh_grid = ideal; if (resc != 0.0) sh_resc = resc; }
      __syncthreads();
    }
    for (int e = tid; e < n; e += blockDim.x) {
      p.gun[e] = p.g[e];
      if (compute_scale) p.scale[e] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(p.Hm[p.row_off[e] + e - p.row_v0[e]])) : 1.0;
    }
    __syncthreads();
    for (int e = tid; e < n; e += blockDim.x) {
      double* row = p.Hm + p.row_off[e];
      const double se = p.scale[e];
      const double* sc = p.scale + p.row_v0[e];
      const int nc = p.row_nc[e];
      for (int c2 = 0; c2 < nc; ++c2) row[c2] *= se * sc[c2];
      p.g[e] *= se;
    }
    __syncthreads();
    KA_T(2);
    return c;
  };
  auto cost_at_candidate = [&](double alpha, bool long_search) -> double {
    stencils += nodes_var;
#ifdef PXR_KA_STREAM_PROBE
    if constexpr (C % KA_SCH == 0 && sizeof(ST) <= 4) {
      if (use_stream && long_search) {
        KA_T(7);
        const double c = fsimd ? ka_probe_stream<ST, C, true>(scall, alpha) : ka_probe_stream<ST, C, false>(scall, alpha);
        KA_T(3);
        return c;
      }
    }
#endif
    ka_plus(a, p, alpha);
    KA_T(7);
    ka_nodes<ST, C, false>(a, p, a.kp_cand, fsimd, true);
    KA_T(3);
    double c = ka_terms<C, false>(a, p, sh4);
    KA_T(4);
    if (grouped) { double z1 = 0.0, z2 = 0.0, z3 = 0.0; gsum4(c, z1, z2, z3); }      // the GROUP's cost at the candidate
    return c;
  };

  if (!grouped && (n == 0 || (p.ne1 == p.ne0 && p.nu1 == p.nu0))) {       // (a member of a group goes through the loop with its siblings)
    ka_nodes<ST, C, false>(a, p, a.v.d_kp, fsimd);
    const double c = ka_terms<C, false>(a, p, sh4);
    sm.initial_cost = sm.final_cost = c; sm.termination = PXR_TERM_CONVERGENCE;
    if (tid == 0) { a.summaries[prob] = sm; if (DET) a.prob_done[prob] = 1; }
    return;
  }
  // A sub-problem that an earlier launch of this solve left with KA_TERM_RESCALE after an ACCEPTED step goes on where it stopped:
  // its Jacobi scaling (fixed at the first linearisation of the solve, [upstream]), radius, decrease factor and counts were
  // parked in prob_state / the global scale vector (it used to start over with a fresh trust region).
  const double* const pst = a.prob_state + (size_t)prob * KA_STATE;
  const bool resumed = pst[0] != 0.0;
  const bool resumed_full = resumed && pst[11] != 0.0;      // parked by phase 1 of the two-phase launch WITH its linearisation
  if (resumed) {
    const double* scale_g = a.vec + 2 * vstride + vb;
    for (int e = tid; e < n; e += blockDim.x) p.scale[e] = scale_g[e];
    if (resumed_full) {
      // H, g (scaled and unscaled) and the damping diagonal come back from global memory: nothing is evaluated again (the
      // descriptors of the constant keypoints, which the probes read, are still in the scratch)
      const double* Hg = a.Hbuf + a.prob_h_ptr[prob];
      const double *g_g = a.vec + 0 * vstride + vb, *gun_g = a.vec + 1 * vstride + vb, *diag_g = a.vec + 3 * vstride + vb;
      if (p.Hm != Hg) for (int e = tid; e < hsz; e += blockDim.x) p.Hm[e] = Hg[e];
      if (p.g != g_g) for (int e = tid; e < n; e += blockDim.x) { p.g[e] = g_g[e]; p.gun[e] = gun_g[e]; p.diag[e] = diag_g[e]; }
      if (DET && tid == 0) sh_grid = pst[9];
    }
    __syncthreads();
  }
  double cost = resumed_full ? uniform_f64(pst[8]) : linearize(!resumed);
  double cost_loc = cost;              // this member's part of the group's cost (what its summary reports: the host adds them up)
  sm.initial_cost = cost_loc;
  bool resc = DET && sh_resc != 0.0, infeasible = !inf.feasible;
  if (grouped) {                       // one Ceres problem: its cost, and its members' flags, are the group's
    double f_resc = resc ? 1.0 : 0.0, f_inf = infeasible ? 1.0 : 0.0, z = 0.0;
    gsum4(cost, f_resc, f_inf, z);
    resc = DET && f_resc > 0.0; infeasible = f_inf > 0.0;
  }
  if (resc) {                          // the grid does not fit this sub-problem (or a sibling): nothing was changed, the host launches again
    sm.final_cost = cost_loc; sm.termination = KA_TERM_RESCALE; sm.linear_iterations = stencils;
    if (tid == 0) { a.summaries[prob] = sm; a.prob_scale[prob] = sh_resc != 0.0 ? sh_resc : sh_grid; }
    return;
  }
  if (infeasible || !isfinite(cost)) {      // [upstream] Program::IsFeasible fails / the initial evaluation fails
                                            // (non-finite input): FAILURE, parameters untouched
    sm.final_cost = cost_loc; sm.termination = PXR_TERM_FAILURE;
    if (tid == 0) { a.summaries[prob] = sm; if (DET) a.prob_done[prob] = 1; }
    return;
  }
  double radius = opt.initial_radius, decrease_factor = 2.0;
  int invalid = 0;
  bool reuse_diag = false;
  if (resumed) {
    radius = uniform_f64(pst[1]); sm.iterations = (int)pst[2]; sm.num_successful = (int)pst[3]; sm.initial_cost = pst[4]; stencils += (int64_t)pst[5];
    decrease_factor = uniform_f64(pst[6]); invalid = (int)pst[7];
    reuse_diag = resumed_full && pst[10] != 0.0;                      // (not FULL: the damping is formed again from the same H)
  }
  const int park_at = sm.iterations + a.park_iters;
  while (true) {
    if (phase == 1 && sm.iterations >= park_at && sm.iterations < opt.max_iterations && !(radius < opt.min_radius)) {
      // phase 1 of the two-phase launch: one LM iteration done, the sub-problem goes on in phase 2 -- those that already sit on a
      // bound (the ones that will run the twenty-probe line searches) first
      sm.termination = KA_TERM_PARKED;
      double* scale_g = a.vec + 2 * vstride + vb;
      int on_bound = 0;
      for (int64_t i = p.np0 + tid; i < p.np1; i += blockDim.x) {
        const int64_t node = i - p.np0 < KA_NODE_CACHE ? sh_nodes[i - p.np0].id : a.v.d_prob_nodes[i];
        const int v = a.var_of_node[node];
        if (v < 0) continue;
#pragma unroll
        for (int c = 0; c < 2; ++c) { const double x = a.v.d_kp[2 * node + c]; if (x == p.lo[v + c] || x == p.hi[v + c]) on_bound = 1; }
      }
      for (int e = tid; e < n; e += blockDim.x) scale_g[e] = p.scale[e];
      {   // the linearisation itself: phase 2 takes it back instead of evaluating the same point again
        double* Hg = a.Hbuf + a.prob_h_ptr[prob];
        double *g_g = a.vec + 0 * vstride + vb, *gun_g = a.vec + 1 * vstride + vb, *diag_g = a.vec + 3 * vstride + vb;
        if (p.Hm != Hg) for (int e = tid; e < hsz; e += blockDim.x) Hg[e] = p.Hm[e];
        if (p.g != g_g) for (int e = tid; e < n; e += blockDim.x) { g_g[e] = p.g[e]; gun_g[e] = p.gun[e]; diag_g[e] = p.diag[e]; }
      }
      on_bound = __syncthreads_or(on_bound);
      if (tid == 0) {
        double* st = a.prob_state + (size_t)prob * KA_STATE;
        st[0] = 1.0; st[1] = radius; st[2] = (double)sm.iterations; st[3] = (double)sm.num_successful; st[4] = sm.initial_cost; st[5] = (double)stencils;
        st[6] = decrease_factor; st[7] = (double)invalid; st[8] = cost; st[9] = DET ? sh_grid : 0.0; st[10] = reuse_diag ? 1.0 : 0.0; st[11] = 1.0;
        a.prob_heavy[prob] = on_bound ? 1 : 0;
      }
      break;
    }
    if (sm.iterations >= opt.max_iterations) { sm.termination = PXR_TERM_NO_CONVERGENCE; break; }
    if (radius < opt.min_radius) { sm.termination = PXR_TERM_CONVERGENCE; break; }
    ++sm.iterations;
    if (!reuse_diag)
      for (int e = tid; e < n; e += blockDim.x)
        p.diag[e] = fmin(fmax(p.Hm[p.row_off[e] + e - p.row_v0[e]], opt.min_lm_diagonal), opt.max_lm_diagonal);
    if (tid == 0) sh_ok = 1;
    __syncthreads();
    for (int e = tid; e < n; e += blockDim.x) {
      const int off = p.row_off[e], nc = p.row_nc[e], dcol = e - p.row_v0[e];
      for (int c2 = 0; c2 < nc; ++c2) A[off + c2] = p.Hm[off + c2] + (c2 == dcol ? p.diag[e] / radius : 0.0);
      p.step[e] = -p.g[e];
    }
    __syncthreads();
    KA_T(8);
    // per-component Cholesky: threads take the 2x2 blocks, wavefronts the blocks up to 64, the
    // workgroup anything larger
    for (int c = tid; c < p.ncomp; c += blockDim.x) {
      const int v0 = p.comp_v0[c];
      if (p.row_nc[v0] == 2 && !ka_chol_2x2(A + p.row_off[v0], p.step + v0)) sh_ok = 0;
    }
    if (maxnc > 2) {
      for (int c = tid >> 6; c < p.ncomp; c += KA_NT / 64) {
        const int v0 = p.comp_v0[c], nc = p.row_nc[v0];
        if (nc > 2 && nc <= KA_NREG) { if (!ka_chol_wave_reg(A + p.row_off[v0], nc, p.step + v0)) sh_ok = 0; }
        else if (nc > 2 && nc <= 64 && !ka_chol_wave(A + p.row_off[v0], nc, p.step + v0)) sh_ok = 0;
      }
    }
    __syncthreads();
    if (maxnc > 64) {
      for (int c = 0; c < p.ncomp; ++c) {
        const int v0 = p.comp_v0[c], nc = p.row_nc[v0];
        if (nc > 64 && !ka_chol_block(A + p.row_off[v0], nc, p.step + v0) && tid == 0) sh_ok = 0;
      }
      __syncthreads();
    }
    KA_T(5);
    bool ok = sh_ok != 0;
    // model cost change = -d.g - 0.5 d.H.d
    double part = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
      const double* row = p.Hm + p.row_off[i];
      const double* sp = p.step + p.row_v0[i];
      const int nc = p.row_nc[i];
      double hr = 0.0;
      for (int j = 0; j < nc; ++j) hr = fma(row[j], sp[j], hr);
      part += -p.step[i] * p.g[i] - 0.5 * p.step[i] * hr;
      if (!isfinite(p.step[i])) part = NAN;
    }
    double model_cost_change = block_sum(part, sh4);
    double g0 = 0.0;
    if (grouped) {
      // the step of a group is valid / invalid as a whole: the directional derivative is formed before the verdict so that the
      // three scalars travel in ONE sum (a member whose factorisation failed contributes NaN-free zeros and the flag)
      double g0p = 0.0;
      for (int e = tid; e < n; e += blockDim.x) {
        const double dl = p.step[e] * p.scale[e];
        p.delta[e] = dl;
        g0p += p.gun[e] * dl;
      }
      g0 = block_sum(g0p, sh4);
      double bad = ok ? 0.0 : 1.0, z = 0.0;
      if (!ok) { model_cost_change = 0.0; g0 = 0.0; }
      gsum4(model_cost_change, g0, bad, z);
      ok = bad == 0.0;
    }
    if (!(model_cost_change > 0.0)) ok = false;
    if (!ok) {
      if (++invalid >= opt.max_consecutive_invalid_steps) { sm.termination = PXR_TERM_FAILURE; break; }
      radius = uniform_f64(radius * 0.5); reuse_diag = true;
      continue;
    }
    invalid = 0;
    if (!grouped) {
      double g0p = 0.0;
      for (int e = tid; e < n; e += blockDim.x) {
        const double dl = p.step[e] * p.scale[e];
        p.delta[e] = dl;
        g0p += p.gun[e] * dl;
      }
      g0 = block_sum(g0p, sh4);
    }
    KA_T(6);
    // DoLineSearch [upstream trust_region_minimizer.cc]: projected Armijo search along delta.  The
    // accepted probe IS the candidate point P(x + delta): its cost is reused instead of re-evaluated.
    double cand;
    {
      double xc = 1.0, xp = 0.0, fp = 0.0;
      bool have_prev = false, success = false;
      int iters = 0;
#ifdef PXR_KA_STREAM_FROM       // (experiment: the streamed probe only from this probe of a search on -- 6.7 ms, worse than always: 6.1)
      constexpr int kStreamFrom = PXR_KA_STREAM_FROM;
#else
      constexpr int kStreamFrom = 0;
#endif
      double fc = cost_at_candidate(xc, kStreamFrom <= 0);
      const double f_full = fc;
      while (true) {
#ifdef PXR_KA_TRACE_LS      // tools/ka_phase_probe.sh: the line search of the reporting sub-problem, probe by probe
        if (blockIdx.x == PXR_KA_PROFILE_BLOCK && tid == 0)
          printf("[ka ls] it %d probe %d  x %.6e  f %.12e  cost %.12e  g0 %.6e  radius %.4e\n", sm.iterations, iters, xc, fc, cost, g0, radius);
#endif
        if (isfinite(fc) && fc <= cost + 1e-4 * g0 * xc) { success = true; break; }
        if (++iters >= 20) break;
        double nx;
        if (!isfinite(fc)) nx = fmin(fmax(xc * 0.5, xc * 1e-3), xc * 0.6);
        else nx = ka_interp_step(cost, g0, have_prev, xp, fp, xc, fc, xc * 1e-3, xc * 0.6);
        nx = uniform_f64(nx);
        if (nx < 1e-9) break;
        if (isfinite(fc)) { xp = xc; fp = fc; have_prev = true; }
        xc = nx;
        fc = cost_at_candidate(xc, iters >= kStreamFrom);
      }
      // (the streamed probes do not write the candidate keypoints: once, here)
      if (success) {
        if (xc != 1.0) {
          for (int e = tid; e < n; e += blockDim.x) p.delta[e] *= xc;
          __syncthreads();
          ka_plus(a, p, 1.0);
        } else if (use_stream) {
          ka_plus(a, p, 1.0);
        }
        cand = fc;
      } else {
        cand = f_full;
        if (xc != 1.0 || use_stream) ka_plus(a, p, 1.0);
      }
    }
    KA_T(9);
    double s2 = 0.0, x2 = 0.0;
    for (int64_t i = p.np0 + tid; i < p.np1; i += blockDim.x) {
      const bool cached = i - p.np0 < KA_NODE_CACHE;
      const int64_t node = cached ? sh_nodes[i - p.np0].id : a.v.d_prob_nodes[i];
      if ((cached ? sh_nodes[i - p.np0].v : a.var_of_node[node]) < 0) continue;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double x0 = a.v.d_kp[2 * node + c], d = a.kp_cand[2 * node + c] - x0;
        s2 += d * d; x2 += x0 * x0;
      }
    }
    double s2g = block_sum(s2, sh4), x2g = block_sum(x2, sh4);
    if (grouped) { double z2 = 0.0, z3 = 0.0; gsum4(s2g, x2g, z2, z3); }
    const double step_norm = sqrt(s2g), x_norm = sqrt(x2g);
    KA_T(10);
    if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) { sm.termination = PXR_TERM_CONVERGENCE; break; }
    const double cost_change = cost - cand;
    if (fabs(cost_change) <= opt.function_tolerance * cost) { sm.termination = PXR_TERM_CONVERGENCE; break; }
    const double rel = cost_change / model_cost_change;
    if (rel > opt.min_relative_decrease) {
      for (int64_t i = p.np0 + tid; i < p.np1; i += blockDim.x) {
        const int64_t node = i - p.np0 < KA_NODE_CACHE ? sh_nodes[i - p.np0].id : a.v.d_prob_nodes[i];
        const double nx = a.kp_cand[2 * node], ny = a.kp_cand[2 * node + 1];
        a.v.d_kp[2 * node] = nx; a.v.d_kp[2 * node + 1] = ny;
        if (i - p.np0 < KA_NODE_CACHE) { sh_kp0[2 * (i - p.np0)] = nx; sh_kp0[2 * (i - p.np0) + 1] = ny; }
#ifdef PXR_KA_STREAM_PROBE
        if (use_stream) { strm.kp[2 * (i - p.np0)] = nx; strm.kp[2 * (i - p.np0) + 1] = ny; }
#endif
      }
      __syncthreads();
      cost = linearize(false);
      cost_loc = cost;
      ++sm.num_successful;
      const double tmp = 2.0 * rel - 1.0;
      radius = uniform_f64(fmin(opt.max_radius, radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp)));
      decrease_factor = 2.0; reuse_diag = false;
      bool resc2 = DET && sh_resc != 0.0;
      if (grouped) { double f = resc2 ? 1.0 : 0.0, z2 = 0.0, z3 = 0.0; gsum4(cost, f, z2, z3); resc2 = DET && f > 0.0; }
      if (resc2) {       // the accepted keypoints stay; the LM state is parked and the next launch goes on from here
        sm.termination = KA_TERM_RESCALE;
        double* scale_g = a.vec + 2 * vstride + vb;
        for (int e = tid; e < n; e += blockDim.x) scale_g[e] = p.scale[e];
        if (tid == 0) {
          double* st = a.prob_state + (size_t)prob * KA_STATE;
          st[0] = 1.0; st[1] = radius; st[2] = (double)sm.iterations; st[3] = (double)sm.num_successful; st[4] = sm.initial_cost; st[5] = (double)stencils;
          st[6] = decrease_factor; st[7] = 0.0; st[8] = 0.0; st[9] = 0.0; st[10] = 0.0; st[11] = 0.0;
        }
        break;
      }
    } else {
      radius = uniform_f64(radius / decrease_factor); decrease_factor = uniform_f64(decrease_factor * 2.0); reuse_diag = true;
    }
  }
  sm.final_cost = cost_loc; sm.final_radius = radius; sm.linear_iterations = stencils;
  if (tid == 0) {
    a.summaries[prob] = sm;
    if (DET) {
      if (sm.termination == KA_TERM_RESCALE) a.prob_scale[prob] = sh_resc != 0.0 ? sh_resc : sh_grid;
      else if (sm.termination == KA_TERM_PARKED) a.prob_scale[prob] = grid_used;    // (phase 2 linearises the SAME point again: on the same grid, the same H and g bit for bit)
      else a.prob_done[prob] = 1;
    }
  }
#ifdef PXR_KA_PROFILE
  KA_T(7);
  if (blockIdx.x == PXR_KA_PROFILE_BLOCK && tid == 0)
    printf("[ka profile, workgroup %d, 10 ns ticks] nodes+J %lld  terms+J %lld  scale %lld  nodes(cost) %lld  terms(cost) %lld  "
           "cholesky %lld  model change %lld  other (ka_plus, step choice) %lld  diag+damp %lld  after the line search %lld  step norms %lld | iterations %d unknowns %d stencils %lld | started at tick %lld, ran %lld\n",
           (int)blockIdx.x, ka_prof[0], ka_prof[1], ka_prof[2], ka_prof[3], ka_prof[4], ka_prof[5], ka_prof[6], ka_prof[7], ka_prof[8], ka_prof[9], ka_prof[10], sm.iterations, n,
           (long long)stencils, ka_tstart, ka_t0 - ka_tstart);
#endif
}

// Two entry points over the same body.  The fp16 / fp32 instantiations need ~284 VGPRs unconstrained,
// i.e. one wavefront per SIMD; capped at 256 (two wavefronts per SIMD, ~60 registers spilled to
// scratch) the whole config-2 solve drops from 12.7 to 9.6 ms.  The fp64-storage instantiations keep the
// compiler's own budget: capped, they spill ~950 registers.  (A tighter cap of 3 waves/SIMD produced
// wrong steps on ROCm 7.2 and is not used.)
template <typename ST, int C, bool DET>
#ifndef PXR_KA_WAVES   // debugging knob of tools/ka_occupancy_probe.sh (the 3-waves build that gave wrong steps)
#define PXR_KA_WAVES 2
#endif
__global__ __launch_bounds__(KA_NT) __attribute__((amdgpu_waves_per_eu(PXR_KA_WAVES, PXR_KA_WAVES))) void ka_solve_kernel_occ2(const KaArgs a, const KaInfo* __restrict__ info) {
  extern __shared__ __align__(16) double sh_A[];      // lds_elems doubles (damped blocks) when the sub-problem fits
  int prob = (int)blockIdx.x;
  if (a.order) { prob = a.order[blockIdx.x]; if (prob < 0) return; }          // (phase 2 of the two-launch schedule)
  ka_solve_body<ST, C, DET>(a, info, sh_A, prob, a.phase);
}

// the order of phase 2: the parked sub-problems, those with a keypoint on a bound first; -1 for the rest of the grid
__global__ __launch_bounds__(1024) void ka_order_kernel(const KaArgs a, int* __restrict__ order) {
  __shared__ int cnt;
  const int np = a.v.n_problems;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = threadIdx.x; i < np; i += blockDim.x)
      if (a.summaries[i].termination == KA_TERM_PARKED && (a.prob_heavy[i] != 0) == (pass == 0)) order[atomicAdd(&cnt, 1)] = i;
    __syncthreads();
  }
  for (int i = cnt + (int)threadIdx.x; i < np; i += blockDim.x) order[i] = -1;
}

// ---- the two-phase launch (round 6; an experiment, off by default: see pxr_ka_solve) ----------------------------------------------
// One workgroup per sub-problem and dispatch in index order make the launch as long as its tail: the ~6 % of the sub-problems
// that sit on an active bound run ~125 line-search probes (2.5-3.3 ms each), the others ~5 (0.65-0.9 ms), and the launch ends when
// the LAST-dispatched heavy one does -- 5.3 ms at configs[1] against 4.1 ms of work per resident workgroup.  Which sub-problems are
// heavy shows after ONE LM iteration: a keypoint whose unconstrained step crosses its bound is clamped onto it.  So a PERSISTENT
// grid (as many workgroups as the launch keeps resident) first runs iteration 1 of every sub-problem and parks its LM state, meets
// at a grid barrier whose last arrival lists the parked sub-problems -- those with a keypoint on a bound first --, and then takes
// them from that list: the long chains start at once, the short ones fill the gaps.  Sub-problems are independent and every one
// goes through the same arithmetic as in the one-phase launch (a parked one re-linearises at the point it stopped at: the same
// H and g bit for bit in the deterministic mode), so the results do not depend on the order.
template <typename ST, int C, bool DET>
__global__ __launch_bounds__(KA_NT) __attribute__((amdgpu_waves_per_eu(PXR_KA_WAVES, PXR_KA_WAVES))) void ka_solve_kernel_sched(const KaArgs a, const KaInfo* __restrict__ info) {
  extern __shared__ __align__(16) double sh_A[];
  __shared__ int sh_take, sh_last, sh_cnt[2];
  const int np = a.v.n_problems, tid = threadIdx.x;
  // ONE call site of the solve body for both phases (two inlined copies of it: 531 spilled registers instead of 68, and a
  // launch of 7.1 ms instead of 5.3)
  int phase = 1, next = blockIdx.x, count = 0;
  while (true) {
    int prob;
    if (phase == 1) {
      if (next < np) { prob = next; next += gridDim.x; }
      else {
        // grid barrier (every workgroup of the grid is resident: the host sized it from the occupancy); the last arrival lists
        // the parked sub-problems, those with a keypoint on a bound first
        if (tid == 0) {
          __threadfence();
          sh_last = __hip_atomic_fetch_add(a.sched + 0, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
          sh_cnt[0] = 0; sh_cnt[1] = 0;
        }
        __syncthreads();
        if (sh_last) {
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          for (int pass = 0; pass < 2; ++pass) {
            for (int i = tid; i < np; i += blockDim.x) {
              const bool parked = __hip_atomic_load(&a.summaries[i].termination, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == KA_TERM_PARKED;
              const bool heavy = __hip_atomic_load(a.prob_heavy + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
              if (parked && heavy == (pass == 0)) a.sched_list[atomicAdd(&sh_cnt[0], 1)] = i;
            }
            __syncthreads();
          }
          if (tid == 0) {
            a.sched[2] = sh_cnt[0];
            __threadfence();
            __hip_atomic_store(a.sched + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          }
        } else if (tid == 0) {
          while (__hip_atomic_load(a.sched + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(2);
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        count = __hip_atomic_load(a.sched + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.stream_off == -1) count = 0;          // PXR_KA_PHASE1_ONLY=1 (a timing experiment: phase 1 and the barrier alone)
        phase = 2;
        continue;
      }
    } else {
      if (tid == 0) sh_take = atomicAdd(a.sched + 3, 1);
      __syncthreads();
      const int k = sh_take;
      __syncthreads();
      if (k >= count) break;
      prob = __hip_atomic_load(a.sched_list + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ka_solve_body<ST, C, DET>(a, info, sh_A, __builtin_amdgcn_readfirstlane(prob), phase);
    __syncthreads();
  }
}
template <typename ST, int C, bool DET>
__global__ __launch_bounds__(KA_NT) void ka_solve_kernel(const KaArgs a, const KaInfo* __restrict__ info) {
  extern __shared__ __align__(16) double sh_A[];
  ka_solve_body<ST, C, DET>(a, info, sh_A, (int)blockIdx.x, 0);
}

// ---- per-edge evaluation (parity checks) -------------------------------------------------------------------
template <typename ST, int C>
__global__ __launch_bounds__(256) void ka_eval_kernel(const KaArgs a, bool fsimd, double* __restrict__ cost,
                                                      double* __restrict__ out_r, double* __restrict__ out_J1,
                                                      double* __restrict__ out_J2) {
  constexpr int LPO = KaLay<C>::LPO, CPL = KaLay<C>::CPL, G = 256 / LPO;
  const int grp = threadIdx.x / LPO, sub = threadIdx.x % LPO;
  const int64_t e = (int64_t)blockIdx.x * G + grp;
  if (e >= a.v.n_edges) return;
  const int nn[2] = {a.v.d_edge_src[e], a.v.d_edge_dst[e]};
  double f[2][CPL], gx[2][CPL], gy[2][CPL];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int64_t pi = a.v.d_node_patch[nn[k]];
    const double sx = a.scales[2 * pi], sy = a.scales[2 * pi + 1];
    const double u = a.v.d_kp[2 * (size_t)nn[k]] * sx - 0.5 - (double)a.corners[2 * pi];
    const double v = a.v.d_kp[2 * (size_t)nn[k] + 1] * sy - 0.5 - (double)a.corners[2 * pi + 1];
    const ST* patch = reinterpret_cast<const ST*>(a.arena) + (size_t)pi * a.H * a.W * C;
    double fr[CPL], fc[CPL];
    if constexpr (C >= 8) {
      if (fsimd) interp8<ST, LPO, true, true>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f[k], fr, fc);
      else interp8<ST, LPO, true, false>(patch, a.H, a.W, C, sub, u, v, a.l2_normalize != 0, f[k], fr, fc);
    } else {
      interp_small<ST, C>(patch, a.H, a.W, u, v, a.l2_normalize != 0, f[k], fr, fc);
    }
#pragma unroll
    for (int ch = 0; ch < CPL; ++ch) { gx[k][ch] = fc[ch] * sx; gy[k][ch] = fr[ch] * sy; }
  }
  double s = 0;
#pragma unroll
  for (int ch = 0; ch < CPL; ++ch) { const double r = f[0][ch] - f[1][ch]; s = fma(r, r, s); }
  s = lpo_sum(s, LPO);
  double rho[3];
  loss_eval(a.loss.type, a.loss.a, a.v.d_edge_w[e], s, rho);
  if (sub == 0) cost[e] = 0.5 * rho[0];   // (check_bounds does not fail a KA residual: the functor returns true, featuremetric.h:61)
  if (out_r) {
#pragma unroll
    for (int ch = 0; ch < CPL; ++ch) {
      const size_t o = (size_t)e * C + sub * CPL + ch;
      out_r[o] = f[0][ch] - f[1][ch];
      if (out_J1) { out_J1[2 * o] = gx[0][ch]; out_J1[2 * o + 1] = gy[0][ch]; }
      if (out_J2) { out_J2[2 * o] = -gx[1][ch]; out_J2[2 * o + 1] = -gy[1][ch]; }
    }
  }
}

static int fill_args(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                     const pxr_loss* loss, KaArgs& a) {
  a.v = *view;
  a.arena = arena->d_data; a.corners = arena->d_corners; a.scales = arena->d_scales;
  a.H = arena->H; a.W = arena->W; a.l2_normalize = cfg->l2_normalize; a.float_simd = cfg->use_float_simd; a.check_bounds = cfg->check_bounds; a.loss = *loss;
  return PXR_OK;
}

template <typename T>
struct KaBuf {
  T* p = nullptr;
  int alloc(size_t n) { return hip_check(hipMalloc((void**)&p, sizeof(T) * (n ? n : 1)), "hipMalloc(KA scratch)"); }
  ~KaBuf() { if (p) (void)hipFree(p); }
};

}  // namespace pxr

extern "C" int pxr_ka_eval(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                           const pxr_loss* loss, double* d_cost, double* d_r, double* d_J1, double* d_J2) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && view && cfg && loss && d_cost, "pxr_ka_eval: NULL argument");
  PXR_REQUIRE(!((d_J1 || d_J2) && !d_r), "pxr_ka_eval: Jacobian outputs require d_r");
  if (view->n_edges == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
  KaArgs a{};
  fill_args(ctx, arena, view, cfg, loss, a);
  const bool fs = cfg->use_float_simd != 0;
#define KA_EVAL_LAUNCH(ST, CC)                                                                                   \
  hipLaunchKernelGGL((ka_eval_kernel<ST, CC>), dim3((unsigned)((view->n_edges + (256 / KaLay<CC>::LPO) - 1) / (256 / KaLay<CC>::LPO))), \
                     dim3(256), 0, ctx->stream, a, fs, d_cost, d_r, d_J1, d_J2)
  if (arena->dtype == PXR_F16 && arena->C == 128) KA_EVAL_LAUNCH(_Float16, 128);
  else if (arena->dtype == PXR_F16 && arena->C == 64) KA_EVAL_LAUNCH(_Float16, 64);
  else if (arena->dtype == PXR_F32 && arena->C == 128) KA_EVAL_LAUNCH(float, 128);
  else if (arena->dtype == PXR_F32 && arena->C == 64) KA_EVAL_LAUNCH(float, 64);
  else if (arena->dtype == PXR_F64 && arena->C == 128) KA_EVAL_LAUNCH(double, 128);
  else if (arena->dtype == PXR_F64 && arena->C == 64) KA_EVAL_LAUNCH(double, 64);
  else if (arena->dtype == PXR_F16 && arena->C == 1) KA_EVAL_LAUNCH(_Float16, 1);
  else if (arena->dtype == PXR_F32 && arena->C == 1) KA_EVAL_LAUNCH(float, 1);
  else if (arena->dtype == PXR_F64 && arena->C == 1) KA_EVAL_LAUNCH(double, 1);
  else return set_error(PXR_EUNSUPPORTED, "pxr_ka_eval: CHANNELS=%d not supported (128, 64, 1)", arena->C);
#undef KA_EVAL_LAUNCH
  return hip_check(hipGetLastError(), "ka_eval_kernel launch");
}

extern "C" int pxr_ka_solve(pxr_ctx* ctx, pxr_arena* arena, const pxr_ka_view* view, const pxr_interp_cfg* cfg,
                            const pxr_loss* loss, double bound, const pxr_lm_options* options,
                            pxr_lm_summary* h_summaries, pxr_lm_summary* total) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && view && cfg && loss && options && total, "pxr_ka_solve: NULL argument");
  PXR_REQUIRE(view->n_unary == 0 || (view->d_unary_node && view->d_unary_ref && view->d_prob_unary_ptr && view->d_prob_unary),
              "pxr_ka_solve: n_unary > 0 needs d_unary_node, d_unary_ref, d_prob_unary_ptr, d_prob_unary");
  PXR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int np = view->n_problems;
  memset(total, 0, sizeof(*total));
  if (np == 0) return PXR_OK;
  std::vector<int64_t> node_ptr(np + 1), edge_ptr(np + 1), h_ptr(np + 1, 0);
  PXR_HIP(hipMemcpyAsync(node_ptr.data(), view->d_prob_node_ptr, sizeof(int64_t) * (np + 1), hipMemcpyDeviceToHost, st));
  PXR_HIP(hipMemcpyAsync(edge_ptr.data(), view->d_prob_edge_ptr, sizeof(int64_t) * (np + 1), hipMemcpyDeviceToHost, st));
  PXR_HIP(hipStreamSynchronize(st));
  PXR_REQUIRE(node_ptr[0] == 0 && node_ptr[np] <= view->n_nodes, "pxr_ka_solve: every node belongs to at most one sub-problem");
  for (int i = 0; i < np; ++i) {
    const int64_t nmax = 2 * (node_ptr[i + 1] - node_ptr[i]);
    PXR_REQUIRE(nmax >= 0 && nmax < 46000, "pxr_ka_solve: sub-problem too large (< 23000 nodes each)");
  }
  const auto t0 = std::chrono::steady_clock::now();
  // every scratch array is carved out of the context's grow-only workspaces (no hipMalloc per call
  // once they have reached their high-water mark)
  auto grow = [&](void** buf, size_t* have, size_t want) -> int {
    if (want <= *have) return PXR_OK;
    PXR_HIP(hipStreamSynchronize(st));
    if (*buf) { PXR_HIP(hipFree(*buf)); *buf = nullptr; *have = 0; }
    if (hipMalloc(buf, want) != hipSuccess) {
      // the BA solver's Gram-matrix cache stays on the context between solves (grow-only, 1.4 GB at 1M observations): give it
      // back before giving up
      (void)hipGetLastError();
      if (ctx->d_gram) { (void)hipFree(ctx->d_gram); ctx->d_gram = nullptr; ctx->gram_bytes = 0; }
      if (ctx->d_solve_arena) { (void)hipFree(ctx->d_solve_arena); ctx->d_solve_arena = nullptr; ctx->solve_arena_bytes = 0; }   // ... and the BA solver's buffers
      PXR_HIP(hipMalloc(buf, want));
    }
    *have = want;
    return PXR_OK;
  };
  const size_t nn = (size_t)view->n_nodes;
  size_t off = 0;
  auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_desc = carve(sizeof(double) * nn * 3 * arena->C), o_cand = carve(sizeof(double) * nn * 2);
  const size_t o_vec = carve(sizeof(double) * nn * 2 * 9);
  const size_t o_var = carve(sizeof(int) * nn), o_label = carve(sizeof(int) * nn), o_ipos = carve(sizeof(int) * nn * 4);
  const size_t o_irow = carve(sizeof(int) * nn * 6), o_comp = carve(sizeof(int) * nn), o_used = carve(nn);
  const size_t o_hptr = carve(sizeof(int64_t) * (np + 1)), o_sum = carve(sizeof(pxr_lm_summary) * np);
  const size_t o_info = carve(sizeof(KaInfo) * np);
  const size_t o_pscale = carve(sizeof(double) * np), o_pdone = carve(np), o_slot = carve(sizeof(int) * nn);
  // label groups spanning several workgroups (pxr_ka_view.d_prob_group)
  std::vector<int> grp_first, grp_size;
  int grp_max = 1;
  if (view->d_prob_group) {
    std::vector<int32_t> gid(np);
    PXR_HIP(hipMemcpyAsync(gid.data(), view->d_prob_group, sizeof(int32_t) * np, hipMemcpyDeviceToHost, st));
    PXR_HIP(hipStreamSynchronize(st));
    grp_first.resize(np); grp_size.resize(np);
    for (int i = 0; i < np;) {
      int j = i;
      while (j < np && gid[j] == gid[i]) ++j;
      PXR_REQUIRE(j == np || gid[j] > gid[i], "pxr_ka_solve: d_prob_group must be non-decreasing (the chunks of a group are consecutive)");
      for (int k = i; k < j; ++k) { grp_first[k] = i; grp_size[k] = j - i; }
      grp_max = std::max(grp_max, j - i);
      i = j;
    }
  }
  const size_t o_pstate = carve(sizeof(double) * KA_STATE * np);
  const size_t o_sched = carve(sizeof(int) * 4), o_slist = carve(sizeof(int) * np), o_heavy = carve(np);
  const size_t o_gfirst = carve(sizeof(int) * np), o_gsize = carve(sizeof(int) * np);
  const size_t o_gpart = carve(sizeof(double) * 2 * 4 * np), o_gcnt = carve(sizeof(unsigned) * KA_GRP_CNT_STRIDE * np);
  if (int rc = grow(&ctx->d_workspace, &ctx->workspace_bytes, off)) return rc;
  char* ws = static_cast<char*>(ctx->d_workspace);
  int64_t* d_hptr = (int64_t*)(ws + o_hptr);
  pxr_lm_summary* d_sum = (pxr_lm_summary*)(ws + o_sum);
  KaInfo* d_info = (KaInfo*)(ws + o_info);
  PXR_HIP(hipMemsetAsync(ws + o_used, 0, nn, st));
  KaArgs a{};
  fill_args(ctx, arena, view, cfg, loss, a);
  a.bound = bound; a.opt = *options;
  // deterministic mode: H and g of a sub-problem in fixed point; the start grid 2^-38 suits unit-norm 128-channel descriptors
  // (bound = max(trace, sqrt(2 trace cost)) ~ 1e4 .. 1e5 at configs[1]), the kernel adapts it per sub-problem and linearisation
  a.det_scale = ctx->deterministic ? 274877906944.0 : 0.0;      // 2^38 (see the grid rule in ka_solve_body)
  a.prob_scale = (double*)(ws + o_pscale); a.prob_done = (uint8_t*)(ws + o_pdone);
  a.prob_state = (double*)(ws + o_pstate);
  PXR_HIP(hipMemsetAsync(a.prob_state, 0, sizeof(double) * KA_STATE * np, st));
  a.sched = (int*)(ws + o_sched); a.sched_list = (int*)(ws + o_slist); a.prob_heavy = (uint8_t*)(ws + o_heavy);
  if (a.det_scale != 0.0) {
    std::vector<double> init((size_t)np, a.det_scale);
    PXR_HIP(hipMemcpyAsync(a.prob_scale, init.data(), sizeof(double) * np, hipMemcpyHostToDevice, st));
    PXR_HIP(hipMemsetAsync(a.prob_done, 0, np, st));
    PXR_HIP(hipStreamSynchronize(st));       // (`init` leaves scope)
  }
  a.desc = (double*)(ws + o_desc); a.kp_cand = (double*)(ws + o_cand); a.vec = (double*)(ws + o_vec);
  a.var_of_node = (int*)(ws + o_var); a.label = (int*)(ws + o_label); a.ipos = (int*)(ws + o_ipos);
  a.irow = (int*)(ws + o_irow); a.comp_v0 = (int*)(ws + o_comp); a.used = (uint8_t*)(ws + o_used);
  a.prob_h_ptr = d_hptr; a.summaries = d_sum; a.slot_of_node = (int*)(ws + o_slot);
  if (grp_max > 1) {
    a.grp_first = (const int*)(ws + o_gfirst); a.grp_size = (const int*)(ws + o_gsize);
    a.grp_part = (double*)(ws + o_gpart); a.grp_cnt = (unsigned*)(ws + o_gcnt);
    PXR_HIP(hipMemcpyAsync(ws + o_gfirst, grp_first.data(), sizeof(int) * np, hipMemcpyHostToDevice, st));
    PXR_HIP(hipMemcpyAsync(ws + o_gsize, grp_size.data(), sizeof(int) * np, hipMemcpyHostToDevice, st));
    PXR_HIP(hipStreamSynchronize(st));
  }
  // 1. components, unknown layout, bounds; the block sizes come back to size the matrices exactly
  hipLaunchKernelGGL(ka_setup_kernel, dim3(np), dim3(KA_NT), 0, st, a, d_info);
  PXR_HIP(hipGetLastError());
  std::vector<KaInfo> infos(np);
  PXR_HIP(hipMemcpyAsync(infos.data(), d_info, sizeof(KaInfo) * np, hipMemcpyDeviceToHost, st));
  PXR_HIP(hipStreamSynchronize(st));
  int64_t largest = 1;
  for (int i = 0; i < np; ++i) {
    h_ptr[i + 1] = h_ptr[i] + infos[i].hsz;
    largest = std::max<int64_t>(largest, infos[i].hsz);
  }
  // LDS for the damped blocks: the largest block-diagonal system, capped at half a CU's LDS; sub-problems beyond the cap keep their damped blocks in global memory (L2)
  const int lds_elems = (int)std::min<int64_t>(largest, (int64_t)KA_NLDS * KA_NLDS);
  const bool need_aglob = largest > lds_elems;
  const size_t hbytes = (sizeof(double) * (size_t)h_ptr[np] + 255) & ~(size_t)255;
  if (int rc = grow(&ctx->d_workspace_mat, &ctx->workspace_mat_bytes, hbytes * (need_aglob ? 2 : 1) + 256)) return rc;
  a.Hbuf = (double*)ctx->d_workspace_mat;
  a.Abuf = need_aglob ? (double*)((char*)ctx->d_workspace_mat + hbytes) : a.Hbuf;
  a.lds_elems = lds_elems;
  // the whole LM state in LDS when EVERY sub-problem fits next to the static metadata caches at two workgroups per CU
  int n_max = 1;
  for (int i = 0; i < np; ++i) n_max = std::max(n_max, infos[i].n);
  n_max = (n_max + 1) & ~1;                                           // the int arrays start 8-byte aligned
  const size_t state_bytes = sizeof(double) * ((size_t)2 * lds_elems + (size_t)8 * n_max) + sizeof(int) * (size_t)4 * n_max;
  a.lds_state_n = (!need_aglob && state_bytes <= (size_t)56 * 1024) ? n_max : 0;
  PXR_HIP(hipMemcpyAsync(d_hptr, h_ptr.data(), sizeof(int64_t) * (np + 1), hipMemcpyHostToDevice, st));
  size_t shmem = a.lds_state_n > 0 ? state_bytes : sizeof(double) * (size_t)lds_elems;
  // the streamed line-search probe (ka_probe_stream): L2-normalised 64 / 128-channel descriptors in fp16 / fp32 storage, no unary
  // terms, every sub-problem of at most 64 nodes with all its residual blocks in the LDS cache, and its LDS region next to the
  // LM state inside the two-workgroups-per-CU budget; PXR_KA_STREAM=0 is the A/B knob
  a.stream_slots = 0; a.stream_off = 0;
#ifdef PXR_KA_STREAM_PROBE
  {
    int64_t nodes_max = 0, edges_max = 0;
    for (int i = 0; i < np; ++i) {
      nodes_max = std::max(nodes_max, node_ptr[i + 1] - node_ptr[i]);
      edges_max = std::max(edges_max, edge_ptr[i + 1] - edge_ptr[i]);
    }
    const char* knob = getenv("PXR_KA_STREAM");
    const size_t stream_bytes = sizeof(double) * (size_t)nodes_max * (KA_SROW + 4);
    const size_t stream_at = (state_bytes + 15) & ~(size_t)15;
    if (!(knob && knob[0] == '0') && a.lds_state_n > 0 && view->n_unary == 0 && cfg->l2_normalize && (arena->C == 128 || arena->C == 64) &&
        (arena->dtype == PXR_F16 || arena->dtype == PXR_F32) && nodes_max >= 1 && nodes_max <= KA_STREAM_MAX && edges_max <= KA_EDGE_CACHE &&
        stream_at + stream_bytes <= (size_t)56 * 1024) {
      a.stream_slots = (int)nodes_max; a.stream_off = (int)(stream_at / sizeof(double));
      shmem = stream_at + stream_bytes;
    }
  }
#else
  (void)edge_ptr;
#endif
  // the two-phase launch (ka_solve_kernel_sched): an EXPERIMENT, off unless PXR_KA_TWO_PHASE=1 (profiles/r6_ka_schedule.txt: the
  // dispatch ORDER is worth 1.8 ms of the 5.3 -- 3.4 ms with the heaviest sub-problems first -- but which ones are heavy does
  // not show after one LM iteration, and the persistent grid costs 0.6 ms in spilled registers and barrier time: 5.55 ms).  Needs
  // more sub-problems than the launch keeps resident, no label groups that span workgroups, the fp16 / fp32 CNN-feature kernels.
  // The two-LAUNCH schedule (round 6, the default when there are more sub-problems than resident workgroups): the launch is as
  // long as its tail -- the ~7 % of the sub-problems that run twenty-probe line searches against an active bound take 2.5-3.3 ms,
  // the others 0.65-0.9, and dispatch in index order starts some of the long ones last (with the heaviest first the same kernel
  // needs 3.4 instead of 5.3 ms at configs[1]: profiles/r6_ka_schedule.txt).  Which ones are heavy shows after TWO LM iterations
  // (a keypoint clamped onto its bound: 131 of 137 caught).  So: launch 1 runs two iterations of every sub-problem and parks the
  // complete LM state (nothing is evaluated twice), a one-workgroup kernel lists the parked ones -- on a bound first --, launch 2
  // resumes them in that order.  Same arithmetic per sub-problem: identical bits (tests/test_ka_gpu.py::test_two_phase_launch_*).
  // 5.36 -> 4.97 ms at configs[1].  PXR_KA_TWO_LAUNCH=0 switches it off, =N parks after N iterations.
  const char* two_launch_knob = getenv("PXR_KA_TWO_LAUNCH");
  const int park_iters = two_launch_knob && atoi(two_launch_knob) > 0 ? atoi(two_launch_knob) : 2;
  int resident_hint = 2 * ctx->num_cus;
  if (const char* e = getenv("PXR_KA_TWO_PHASE_RESIDENT")) resident_hint = std::max(1, atoi(e));
  const bool two_launch = !(two_launch_knob && two_launch_knob[0] == '0') && grp_max <= 1 && np > resident_hint;
  const char* two_phase_knob = getenv("PXR_KA_TWO_PHASE");
  const bool two_phase_wanted = two_phase_knob && two_phase_knob[0] == '1' && grp_max <= 1;
#define KA_SOLVE_LAUNCH(KERNEL, SCHED, ST, CC)                                                                    \
  do {                                                                                                       \
    void (*kfn)(const KaArgs, const KaInfo*) = a.det_scale != 0.0 ? KERNEL<ST, CC, true> : KERNEL<ST, CC, false>; \
    unsigned grid = (unsigned)np;                                                                            \
    void (*sfn)(const KaArgs, const KaInfo*) = a.det_scale != 0.0 ? SCHED<ST, CC, true> : SCHED<ST, CC, false>;   \
    if (two_phase_wanted && sfn != kfn) {                                                                    \
      PXR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
      int per_cu = 0;                                                                                        \
      PXR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(sfn), KA_NT, shmem)); \
      int resident = per_cu * ctx->num_cus;                                                                  \
      if (const char* e = getenv("PXR_KA_TWO_PHASE_RESIDENT")) resident = std::min(resident, std::max(1, atoi(e)));   /* tests: a small grid */ \
      if (getenv("PXR_VERBOSE")) fprintf(stderr, "[pxr_ka_solve] two-phase launch: %d workgroups per CU x %d CUs resident, %d sub-problems, %zu B of dynamic LDS\n", per_cu, ctx->num_cus, np, (size_t)shmem); \
      if (resident >= 1 && np > resident) {        /* (everything resident at once: no tail to schedule away) */ \
        kfn = sfn; grid = (unsigned)resident;                                                                \
        if (getenv("PXR_KA_PHASE1_ONLY")) a.stream_off = -1;                                                 \
        PXR_HIP(hipMemsetAsync(a.sched, 0, sizeof(int) * 4, st));                                            \
        PXR_HIP(hipMemsetAsync(a.prob_heavy, 0, np, st));                                                    \
      }                                                                                                      \
    }                                                                                                        \
    PXR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                          \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));                    \
    if (grp_max > 1) {   /* every member of a group must be resident at the same time (ka_group_sum4 spins) */ \
      int per_cu = 0;                                                                                        \
      PXR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kfn), KA_NT, shmem)); \
      const int resident = per_cu * ctx->num_cus;                                                            \
      if (grp_max > resident * 7 / 8)                                                                        \
        return set_error(PXR_EUNSUPPORTED, "pxr_ka_solve: a label group of %d chunks exceeds what one launch keeps resident (%d workgroups): use smaller groups", grp_max, resident); \
      PXR_HIP(hipMemsetAsync(a.grp_cnt, 0, sizeof(unsigned) * KA_GRP_CNT_STRIDE * np, st));                  \
    }                                                                                                        \
    PXR_HIP(hipEventRecord(ctx->ev_start, st));                                                              \
    if (two_launch && kfn != sfn && sfn != (void (*)(const KaArgs, const KaInfo*))nullptr && (void*)SCHED<ST, CC, true> != (void*)KERNEL<ST, CC, true>) { \
      /* two launches of the plain kernel: every sub-problem runs `park_iters` LM iterations and parks its complete LM state, */ \
      /* a one-workgroup kernel lists the parked ones (those on a bound first), the second launch resumes them in that order  */ \
      PXR_HIP(hipMemsetAsync(a.prob_heavy, 0, np, st));                                                      \
      KaArgs a1 = a; a1.phase = 1; a1.park_iters = park_iters;                                               \
      hipLaunchKernelGGL(kfn, dim3(grid), dim3(KA_NT), shmem, st, a1, d_info);                               \
      hipLaunchKernelGGL(ka_order_kernel, dim3(1), dim3(1024), 0, st, a, a.sched_list);                      \
      KaArgs a2 = a; a2.phase = 2; a2.order = a.sched_list;                                                  \
      hipLaunchKernelGGL(kfn, dim3(grid), dim3(KA_NT), shmem, st, a2, d_info);                               \
    } else                                                                                                   \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(KA_NT), shmem, st, a, d_info);                                  \
    PXR_HIP(hipEventRecord(ctx->ev_stop, st));                                                               \
  } while (0)
  std::vector<pxr_lm_summary> sums(np), pass(np);
  double kernel_ms = 0.0;
  // One launch solves every sub-problem.  Deterministic mode: a sub-problem whose fixed-point grid does not fit (its overflow
  // guard; raw un-normalised features) stops with KA_TERM_RESCALE and its new grid in prob_scale -- launch again: finished
  // sub-problems return at once, the others go on from their current keypoints.  (Never more than one launch with unit-norm
  // descriptors.)
  for (int launch = 0; launch < 8; ++launch) {
    if (arena->dtype == PXR_F16 && arena->C == 128) KA_SOLVE_LAUNCH(ka_solve_kernel_occ2, ka_solve_kernel_sched, _Float16, 128);
    else if (arena->dtype == PXR_F16 && arena->C == 64) KA_SOLVE_LAUNCH(ka_solve_kernel_occ2, ka_solve_kernel_sched, _Float16, 64);
    else if (arena->dtype == PXR_F32 && arena->C == 128) KA_SOLVE_LAUNCH(ka_solve_kernel_occ2, ka_solve_kernel_sched, float, 128);
    else if (arena->dtype == PXR_F64 && arena->C == 128) KA_SOLVE_LAUNCH(ka_solve_kernel, ka_solve_kernel, double, 128);
    else if (arena->dtype == PXR_F32 && arena->C == 64) KA_SOLVE_LAUNCH(ka_solve_kernel_occ2, ka_solve_kernel_sched, float, 64);
    else if (arena->dtype == PXR_F64 && arena->C == 64) KA_SOLVE_LAUNCH(ka_solve_kernel, ka_solve_kernel, double, 64);
    else if (arena->dtype == PXR_F16 && arena->C == 1) KA_SOLVE_LAUNCH(ka_solve_kernel, ka_solve_kernel, _Float16, 1);
    else if (arena->dtype == PXR_F32 && arena->C == 1) KA_SOLVE_LAUNCH(ka_solve_kernel, ka_solve_kernel, float, 1);
    else if (arena->dtype == PXR_F64 && arena->C == 1) KA_SOLVE_LAUNCH(ka_solve_kernel, ka_solve_kernel, double, 1);
    else return set_error(PXR_EUNSUPPORTED, "pxr_ka_solve: CHANNELS=%d not supported (128, 64, 1)", arena->C);
    PXR_HIP(hipGetLastError());
    PXR_HIP(hipMemcpyAsync(pass.data(), d_sum, sizeof(pxr_lm_summary) * np, hipMemcpyDeviceToHost, st));
    PXR_HIP(hipStreamSynchronize(st));
    float kms = 0.f;
    PXR_HIP(hipEventElapsedTime(&kms, ctx->ev_start, ctx->ev_stop));
    kernel_ms += (double)kms;
    bool again = false;
    for (int i = 0; i < np; ++i) {
      if (launch == 0) { sums[i] = pass[i]; }
      else if (sums[i].termination == KA_TERM_RESCALE) {        // this launch continued sub-problem i
        const pxr_lm_summary before = sums[i];
        sums[i] = pass[i];
        // (a sub-problem that had accepted steps resumed from its parked state and reports the running totals itself; one
        //  that stopped at its very first linearisation had done nothing but count its stencils)
        if (before.num_successful == 0) sums[i].linear_iterations += before.linear_iterations;
      }
      again = again || sums[i].termination == KA_TERM_RESCALE;
    }
    if (!again) break;
    PXR_REQUIRE(launch < 7, "pxr_ka_solve: the fixed-point grid of the deterministic mode could not be fitted (non-finite features?)");
  }
#undef KA_SOLVE_LAUNCH
  total->accumulation = a.det_scale != 0.0 ? 1 : 0;
  total->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  total->setup_ms = total->total_ms - kernel_ms;   // everything but the solve kernel itself (workspace, uploads, summaries download)
  total->termination = PXR_TERM_CONVERGENCE;
  for (int i = 0; i < np; ++i) {   // AccumulateSummaries (util/src/statistics.h:131-160)
    total->initial_cost += sums[i].initial_cost; total->final_cost += sums[i].final_cost;
    total->iterations = std::max(total->iterations, sums[i].iterations);
    if (grp_max <= 1 || grp_first[i] == i) total->num_successful += sums[i].num_successful;      // (a group's chunks all report the group's steps)
    total->num_point_unknowns += sums[i].num_camera_unknowns;
    total->linear_iterations += sums[i].linear_iterations;          // node stencils interpolated over the whole solve
    if (sums[i].termination == PXR_TERM_FAILURE) total->termination = PXR_TERM_FAILURE;
    else if (sums[i].termination == PXR_TERM_NO_CONVERGENCE && total->termination != PXR_TERM_FAILURE)
      total->termination = PXR_TERM_NO_CONVERGENCE;
  }
  if (h_summaries) memcpy(h_summaries, sums.data(), sizeof(pxr_lm_summary) * np);
  return PXR_OK;
}

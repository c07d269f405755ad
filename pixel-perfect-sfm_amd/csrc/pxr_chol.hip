// pxr_chol.hip -- dense fp64 Cholesky solve of the reduced camera system on gfx950, written for
// the LM loop of pxr_ba_solve: no host synchronisation, no workspace allocation, a fixed short
// chain of launches (the library path -- rocSOLVER potrf/potrs -- cost 24 ms + 8 ms of host-side
// overhead per call at n = 1593 for ~4 ms of kernels).
//
// Storage: the system is built row-major as the n x (n + 1) matrix [S | rhs] with only the UPPER
// triangle of S filled.  Read column-major with leading dimension ld = n + 1 this is the LOWER
// triangle A(i,j) = a[i + j*ld], i >= j, plus one extra ROW i = n holding the right-hand side:
// a right-looking Cholesky that treats row n like any other sub-diagonal row performs the forward
// substitution L y = b for free (row n of the factor is y^T).
//
// ONE launch per 64-column block step s (k_chol_step), look-ahead built into the launch:
//   panel workgroups   (8 wavefronts; 48 rows each) first apply step s - 1's rank-64 update to THEIR part of block
//                      column s -- the 64 x 64 diagonal tile (every panel workgroup redundantly: no hop between
//                      workgroups on the critical path) and their own rows below it -- on fp64 MFMA accumulators
//                      that never leave the registers, then factor the tile and solve their rows in the same sweep
//                      (pxr_chol_core.h: lane = row pivot chain on v_readlane, rank-8 MFMA updates, two barriers per 8
//                      columns).  The first 64 "rows" are identity rows: they come out as inv(L_ss)^T for the
//                      back-substitution;
//   update workgroups  apply step s - 1's update to the tiles RIGHT of block column s (64 x 64 MFMA tiles), off the
//                      critical path of the step.
// So step s needs only what step s - 1 wrote, the trailing update never delays the next pivot chain, and the chain
// itself is ~300 shader cycles per pivot instead of ~1000 (profiles/r3_chol_*).
//   k_chol_backsolve   L^T x = y in one launch: a workgroup per block column, solution blocks handed on
//                      through flags; x_k = inv(L_kk)^T z_k, z_c -= L(k-block, c-block)^T x_k.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "pxr_chol_core.h"
#include "pxr_internal.h"

namespace pxr {

constexpr int CNB = 64;
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
constexpr int SPAD = CNB + 8;   // LDS row stride of the update tiles: spreads the four k-rows of a step over the banks

// LDS of a panel workgroup: the sweep's buffers + step s - 1's columns of the tile rows (64) and of the workgroup's own
// rows (48), in operand order U[k][row]
struct PanelLds {
  cholcore::FactorLds f;
  double U[CNB][CNB + 16 * cholcore::ROW_WAVES];
  double H[3][4][CNB];    // the three tile blocks wavefront 4 updates on behalf of wavefronts 3, 3, 2 (accumulator order)
};
struct UpdateLds {
  double Pi[CNB][SPAD];   // [m][row of the row tile]
  double Pj[CNB][SPAD];   // [m][row of the column tile]
};
// ONE workgroup per CU: a panel workgroup that shares its CU with an update workgroup shares its MFMA and LDS pipes, and
// its pivot chain is the critical path of the step -- asking for more than half of the CU's 160 KiB keeps a second
// workgroup out; when a step has more than 256 workgroups the surplus update tiles run as a short second round
constexpr size_t kStepLdsNeed = sizeof(PanelLds) > sizeof(UpdateLds) ? sizeof(PanelLds) : sizeof(UpdateLds);
constexpr size_t kStepLds = kStepLdsNeed > 96 * 1024 ? kStepLdsNeed : 96 * 1024;

// trailing tile (ti, tj) of the matrix below / right of block column s - 1:  A(ti, tj) -= P_ti P_tj^T with
// P = A(:, 64 (s - 1) .. 64 s) -- fp64 MFMA (v_mfma_f64_16x16x4_f64), four wavefronts, each a 32 x 32 quarter = 2 x 2 MFMA
// blocks, K = 64 in 16 steps.  Operands come from the LDS-staged panel rows (P[m][row]: lane l feeds row l % 16,
// k = l / 16 for A and for B alike).  The product is formed TRANSPOSED (A <- rows of the column tile, B <- rows of the row
// tile) so that the accumulator's lane index (col = lane & 15) runs along the matrix ROWS, which are contiguous in memory:
// C/D layout of the f64 MFMA is col = lane & 15, row = (lane >> 4) + 4 * reg.
__device__ __forceinline__ void update_tile(UpdateLds& lds, double* __restrict__ a, int n_rows, int lda, int k, int r0, int ti, int tj) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < CNB * CNB / 512; ++q) {
    const int e = tid + q * 512, i = e % CNB, m = e / CNB;
    const int gi = r0 + ti * CNB + i, gj = r0 + tj * CNB + i;
    // clamped address + select (a conditional load costs a branch and a wait per element)
    const size_t col = (size_t)(k + m) * lda;
    const double vi = a[(size_t)min(gi, n_rows - 1) + col], vj = a[(size_t)min(gj, n_rows - 1) + col];
    lds.Pi[m][i] = gi < n_rows ? vi : 0.0;
    lds.Pj[m][i] = gj < n_rows ? vj : 0.0;
  }
  // the tile itself goes INTO the accumulators before the products (loads in flight behind the staging; a read-modify-write
  // after the MFMAs serialised 16 load -> wait -> store trips per lane)
  const int lane = tid & 63, w = (tid >> 6) & 3, wi = w >> 1, wj = w & 1;
  const int lr = lane & 15, lk = lane >> 4;
  mfma_d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gj = r0 + tj * CNB + 32 * wj + 16 * x + lk + 4 * r;
        const int gi = r0 + ti * CNB + 32 * wi + 16 * y + lr;
        const double c = a[(size_t)min(gi, n_rows - 1) + (size_t)min(gj, n_rows - 1) * lda];     // clamped address + select
        acc[x][y][r] = (gi < n_rows && gj < n_rows && gi >= gj) ? c : 0.0;
      }
  __syncthreads();
  if (tid >= 256) return;
#pragma unroll 4
  for (int s = 0; s < CNB / 4; ++s) {
    const int m = 4 * s + lk;
    const double aj0 = -lds.Pj[m][32 * wj + lr], aj1 = -lds.Pj[m][32 * wj + 16 + lr];     // A: column-tile rows (negated: acc -= P P^T)
    const double bi0 = lds.Pi[m][32 * wi + lr], bi1 = lds.Pi[m][32 * wi + 16 + lr];       // B: row-tile rows
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj0, bi0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj0, bi1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj1, bi0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj1, bi1, acc[1][1], 0, 0, 0);
  }
  // acc[x][y][r] = A[gi][gj] - sum_m P[gj][m] P[gi][m]  with  gj = column-tile row 32 wj + 16 x + (lane >> 4) + 4 r,
  //                                                           gi = row-tile row    32 wi + 16 y + (lane & 15)
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gj = r0 + tj * CNB + 32 * wj + 16 * x + lk + 4 * r;
        const int gi = r0 + ti * CNB + 32 * wi + 16 * y + lr;
        // write-through (sc1) stores: the ~9 MB of updated tiles leave the L2 while the panel workgroups are still busy,
        // instead of as one write-back of dirty lines at the end of the launch (which the next step waits for)
        if (gi < n_rows && gj < n_rows && gi >= gj)
          __hip_atomic_store(&a[(size_t)gi + (size_t)gj * lda], acc[x][y][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
}

// acc[cb] -= U_rows U_cols(cb)^T over the 64 columns of the previous block step, blocks CB0 <= cb < CB1
template <int CB0, int CB1>
__device__ __forceinline__ void rank64_update(cholcore::d4 (&acc)[4], const double (&U)[CNB][CNB + 16 * cholcore::ROW_WAVES], int rrow, int i, int g) {
#pragma unroll 4
  for (int ks = 0; ks < CNB / 4; ++ks) {
    const double br = U[4 * ks + g][rrow];
#pragma unroll
    for (int cb = CB0; cb < CB1; ++cb) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(-U[4 * ks + g][16 * cb + i], br, acc[cb], 0, 0, 0);
  }
}

// block step s: n_rows = rows of the (augmented) matrix, n_cols = columns to factor.  Grid: n_panel panel workgroups, then
// the update workgroups of the tiles (ti, tj), 1 <= tj <= ti, of the matrix from block row s on (s >= 1 only).
__device__ __forceinline__ void chol_step_body(unsigned char* smem, double* __restrict__ a, int n_rows, int n_cols, int lda, int s, int n_panel,
                                               int block, int* __restrict__ info, double* __restrict__ linv_out, double* __restrict__ ldiag_out) {
  using namespace cholcore;
  const int k0 = s * CNB, nb = min(CNB, n_cols - k0);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (block >= n_panel) {                                             // ---- an update workgroup
    int t = block - n_panel, ti = 1;
    while (t >= ti) { t -= ti; ++ti; }                                 // tile t of row ti: (ti, 1 + t), at most ~25 rows
    update_tile(*reinterpret_cast<UpdateLds*>(smem), a, n_rows, lda, k0 - CNB, k0, ti, 1 + t);
    return;
  }
  // ---- a panel workgroup
  PanelLds& lds = *reinterpret_cast<PanelLds*>(smem);
  const int i = lane & 15, g = lane >> 4;
  // the 16-row block this wavefront carries through the sweep: blocks 0..3 are identity rows (-> inv(L_ss)), then the rows
  // below the tile
  const int v = wave - ROW_WAVE0, vb = ROW_WAVES * block + v;
  const int r0 = k0 + nb + 16 * (vb - 4);                              // first matrix row of a real row block
  const bool real_rows = wave >= ROW_WAVE0 && vb >= 4 && r0 < n_rows;
  RowSink sink;
  // L_ss goes to the workspace, NOT over the tile: the other panel workgroups read the tile when they start, and with more
  // panel workgroups than CUs (n > ~12000) the late ones start after workgroup 0 has finished
  sink.l_diag = block == 0 ? ldiag_out + (size_t)s * CNB * CNB : nullptr;
  sink.ld = lda; sink.nb = nb; sink.rows_valid = 16; sink.rows_out = nullptr; sink.mode = RowSink::kNone;
  d4 acc[4];
  if (wave < 4) {
    load_diag_blocks(acc, a + (size_t)k0 + (size_t)k0 * lda, lda, nb, wave, lane);
  } else if (wave >= ROW_WAVE0 && vb < 4) {
    identity_rows(acc, vb, lane);
    sink.mode = RowSink::kInverse; sink.rows_out = linv_out + (size_t)s * CNB * CNB + 16 * vb;
  } else if (real_rows) {
    sink.mode = RowSink::kPanel; sink.rows_out = a + (size_t)r0 + (size_t)k0 * lda; sink.rows_valid = min(16, n_rows - r0);
    load_rows(acc, sink.rows_out, lda, nb, sink.rows_valid, lane);
  }
  if (s > 0) {
    // step s - 1's rank-64 update of this workgroup's part of block column s.  Stage L(tile rows, s-1) and
    // L(own rows, s-1) as U[k][row] (rows contiguous in memory: coalesced), then MFMA from LDS.
    const int kp = k0 - CNB;
    const int first_real = k0 + nb + 16 * (ROW_WAVES * block - 4);   // row of this workgroup's row block v = 0
    // thread t: row t % 128 of the 112 staged rows (tile rows, then this workgroup's 48), columns t / 128 + 4 q -- all 16
    // loads are issued before the first LDS store (a load -> store loop pays the memory latency once per trip)
    {
      const int r = tid & 127, m0 = tid >> 7;
      int gr = -1;
      if (r < CNB) { if (r < nb) gr = k0 + r; }
      else if (r < CNB + 16 * ROW_WAVES) {
        const int vbb = ROW_WAVES * block + ((r - CNB) >> 4);
        if (vbb >= 4) gr = first_real + (r - CNB);
      }
      const bool live = gr >= 0 && gr < n_rows;
      const double* src = a + (size_t)(live ? gr : 0) + (size_t)(kp + m0) * lda;
      double u[CNB / 4];
#pragma unroll
      for (int q = 0; q < CNB / 4; ++q) u[q] = src[(size_t)(4 * q) * lda];
      if (r < CNB + 16 * ROW_WAVES) {
#pragma unroll
        for (int q = 0; q < CNB / 4; ++q) lds.U[m0 + 4 * q][r] = live ? u[q] : 0.0;
      }
    }
    __syncthreads();
    // MFMA balance over the four SIMDs: wavefront w and row wavefront w + 4 share SIMD w; block row 3 (4 blocks) and a row
    // wavefront (4 blocks) would put 128 MFMAs on SIMD 3 against 16 on SIMD 0, so the otherwise idle wavefront 4 (SIMD 0)
    // takes blocks (3,0), (3,1), (2,0) and hands its sums over through LDS: 96 per SIMD at most
    if (wave < 4 || real_rows) {
      const int rrow = wave < 4 ? 16 * wave + i : CNB + 16 * v + i;    // this lane's row of the B (row) operand
      // one clean loop per role (a run-time block range inside ONE loop made the compiler branch around every MFMA)
      if (wave == 0) rank64_update<0, 1>(acc, lds.U, rrow, i, g);
      else if (wave == 1) rank64_update<0, 2>(acc, lds.U, rrow, i, g);
      else if (wave == 2) rank64_update<1, 3>(acc, lds.U, rrow, i, g);
      else if (wave == 3) rank64_update<2, 4>(acc, lds.U, rrow, i, g);
      else rank64_update<0, 4>(acc, lds.U, rrow, i, g);
    } else if (wave == 4) {
      d4 h0 = d4{0.0, 0.0, 0.0, 0.0}, h1 = h0, h2 = h0;
#pragma unroll 4
      for (int ks = 0; ks < CNB / 4; ++ks) {
        const double br3 = lds.U[4 * ks + g][48 + i], br2 = lds.U[4 * ks + g][32 + i];
        const double a0 = -lds.U[4 * ks + g][i], a1 = -lds.U[4 * ks + g][16 + i];
        h0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, br3, h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, br3, h1, 0, 0, 0);
        h2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, br2, h2, 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) { lds.H[0][t][lane] = h0[t]; lds.H[1][t][lane] = h1[t]; lds.H[2][t][lane] = h2[t]; }
    }
    __syncthreads();
    if (wave == 3) {
#pragma unroll
      for (int t = 0; t < 4; ++t) { acc[0][t] += lds.H[0][t][lane]; acc[1][t] += lds.H[1][t][lane]; }
    } else if (wave == 2) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[0][t] += lds.H[2][t][lane];
    }
  }
  const int bad = factor_tile(lds.f, acc, wave, lane, sink);
  if (block == 0 && tid == 0 && bad && bad <= nb) atomicCAS(info, 0, k0 + bad);
}

// one launch per block step (the form of rounds 3-5; PXR_CHOL_STEP_LAUNCHES=1 and the standalone pxr_chol C-ABI keep it)
__global__ __launch_bounds__(512) void k_chol_step(double* __restrict__ a, int n_rows, int n_cols, int lda, int s, int n_panel,
                                                   int* __restrict__ info, double* __restrict__ linv_out, double* __restrict__ ldiag_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  chol_step_body(smem, a, n_rows, n_cols, lda, s, n_panel, (int)blockIdx.x, info, linv_out, ldiag_out);
}

// ALL block steps in ONE launch (round 6; an EXPERIMENT, off by default -- see chol_factor_solve).  The chain of rounds 3-5 -- 25 launches at n = 1593 -- paid ~5 us of dispatch + cache
// maintenance between two dependent kernels, a third of a 17 us step (profiles/r6_lm_timeline.txt).  Here the grid is the
// concatenation of every step's workgroups (panel workgroups first, then the update tiles, step after step); a workgroup of
// step s > 0 waits until ALL workgroups of step s - 1 have counted themselves done (done[s - 1] == their number) -- exactly the
// dependency the stream order expressed.  Dependencies only point to LOWER workgroup ids, which the dispatcher starts first, so
// the spin-waits cannot deadlock even when the grid exceeds the machine (the argument of k_chol_backsolve).  Release: every
// thread's stores, a workgroup barrier, then one thread's __threadfence + atomic add; acquire: one thread's atomic load at agent
// scope (it invalidates the CU's L1), a barrier, then plain loads.
constexpr int kDoneStride = 32;      // one 128-byte line per step counter
__global__ __launch_bounds__(512) void k_chol_all(double* __restrict__ a, int n_rows, int n_cols, int lda, int* __restrict__ info,
                                                  double* __restrict__ linv_out, double* __restrict__ ldiag_out, int* __restrict__ done) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int sh_step[3];
  const int rows_per_wg = 16 * cholcore::ROW_WAVES;
  if (threadIdx.x == 0) {
    int first = 0, s = 0, prev_count = 0;
    for (int k = 0; k < n_cols; ++s, k += CNB) {          // (at most n / 64 trips of integer arithmetic)
      const int nb = min(CNB, n_cols - k), rem = n_rows - k - nb;
      const int n_panel = (CNB + rem + rows_per_wg - 1) / rows_per_wg;
      const int T = (n_rows - k + CNB - 1) / CNB;
      const int count = n_panel + (s > 0 ? T * (T - 1) / 2 : 0);
      if ((int)blockIdx.x < first + count) { sh_step[0] = s; sh_step[1] = (int)blockIdx.x - first; sh_step[2] = n_panel; break; }
      first += count; prev_count = count;
    }
    if (sh_step[0] > 0) {
      // poll RELAXED (an acquire per poll invalidates this XCD's caches every time: with ~200 waiting workgroups that made the
      // whole factorisation 2x slower than the chain of launches), then ONE acquire fence before the data is read
      const int* flag = done + kDoneStride * (sh_step[0] - 1);
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < prev_count) __builtin_amdgcn_s_sleep(2);
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
  }
  __syncthreads();
  const int s = sh_step[0];
  chol_step_body(smem, a, n_rows, n_cols, lda, s, sh_step[2], sh_step[1], info, linv_out, ldiag_out);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    __hip_atomic_fetch_add(done + kDoneStride * s, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// L^T x = y in ONE launch: workgroup b owns block column c = nblk - 1 - b (64 unknowns) and keeps its
// slice z_c of the right-hand side in LDS.  It consumes the solution blocks x_k, k = nblk-1 .. c+1, in
// order as they are published (flag per block, acquire/release through L2):  z_c -= L(k-block, c-block)^T x_k
// (64 x 64 mat-vec on its own tile of the factor), then x_c = inv(L_cc)^T z_c with the stored block
// inverse, publishes it and raises its flag.  Dependencies only point to LOWER workgroup ids, which the
// dispatcher starts first, so the spin-waits cannot deadlock even when the grid exceeds the machine.
// Critical path: nblk x (one 64x64 mat-vec + one flag hop) instead of nblk kernel launches.
__global__ __launch_bounds__(256) void k_chol_backsolve(const double* __restrict__ a, int n, int lda,
                                                        const double* __restrict__ linv, double* x_out,
                                                        int* flags, int nblk) {
  __shared__ double z[CNB], xk[CNB], part[4][CNB];
  const int tid = threadIdx.x;
  const int c = nblk - 1 - (int)blockIdx.x;
  const int c0 = c * CNB, nbc = min(CNB, n - c0);
  const int j = tid & (CNB - 1), sl = tid >> 6;     // column j of the tile, row slice sl (4 x 16 rows)
  if (tid < CNB) z[tid] = (tid < nbc) ? a[(size_t)n + (size_t)(c0 + tid) * lda] : 0.0;
  for (int k = nblk - 1; k > c; --k) {
    const int k0 = k * CNB, nbk = min(CNB, n - k0);
    // issue this tile's loads before waiting for x_k: L(k0 + r, c0 + j), r contiguous in memory
    double lv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = sl * 16 + q;
      lv[q] = (r < nbk && j < nbc) ? a[(size_t)(k0 + r) + (size_t)(c0 + j) * lda] : 0.0;
    }
    if (tid == 0) while (__hip_atomic_load(&flags[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {}
    __syncthreads();
    if (tid < CNB) xk[tid] = (tid < nbk) ? __hip_atomic_load(&x_out[k0 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s = fma(lv[q], xk[sl * 16 + q], s);
    part[sl][j] = s;
    __syncthreads();
    if (tid < CNB) z[tid] -= part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
  }
  __syncthreads();
  // x_c = inv(L_cc)^T z_c:  x[j] = sum_{r >= j} Linv[r][j] z[r]
  const double* Li = linv + (size_t)c * CNB * CNB;
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int r = sl * 16 + q;
    s = fma(Li[r * CNB + j], z[r], s);              // entries above the diagonal / beyond nbc are stored as 0
  }
  part[sl][j] = s;
  __syncthreads();
  if (tid < nbc) __hip_atomic_store(&x_out[c0 + tid], part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid],
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (tid == 0) {
    __threadfence();
    __hip_atomic_store(&flags[c], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// linv_ws: (2 nblk + 1) x 64 x 64 doubles, nblk = ceil(n / 64): [0, nblk) the inverses of the diagonal tiles' factors,
// [nblk] the back-substitution's flags, [nblk + 1, 2 nblk + 1) the diagonal tiles' factors L_ss (column-major 64 x 64; in
// `a` the diagonal tiles keep their INPUT values, see k_chol_step)
size_t chol_workspace_doubles(int n) { return (size_t)(2 * ((n + CNB - 1) / CNB) + 1) * CNB * CNB; }

int chol_factor_solve(hipStream_t st, double* a, int n, int* d_info, double* linv_ws, double* x_out, bool zero_info) {
  const int lda = n + 1, n_rows = n + 1;
  const int nblk = (n + CNB - 1) / CNB;
  // > 64 KiB of dynamic LDS needs the opt-in, per device (one process may hold contexts on several devices)
  if (int rc = hip_check(hipFuncSetAttribute((const void*)k_chol_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStepLds), "LDS size")) return rc;
  if (zero_info) if (int rc = hip_check(hipMemsetAsync(d_info, 0, sizeof(int), st), "memset info")) return rc;
  const int rows_per_wg = 16 * cholcore::ROW_WAVES;
  // the spare 64 x 64 block of the workspace: [0, nblk) the back-substitution's flags, [nblk, 2 nblk) the steps' completion counters
  int* flags = reinterpret_cast<int*>(linv_ws + (size_t)nblk * CNB * CNB);
  const size_t spare_ints = sizeof(double) * CNB * CNB / sizeof(int);
  if ((size_t)nblk > spare_ints) return set_error(PXR_EINVAL, "chol_factor_solve: n = %d too large", n);
  // one launch for all steps needs a 128-byte line per step counter behind the flags: n <= ~15800; beyond, the chain of launches
  const bool one_launch_fits = (size_t)nblk * (1 + kDoneStride) <= spare_ints;
  // MEASURED AND NOT ADOPTED (round 6, n = 1593, profiles/r6_chol_one_launch.txt): one launch for all steps 2.10 ms per LM iteration
  // against 2.00 ms for the chain of 25 launches -- a release at agent scope writes the XCD's L2 back and an acquire invalidates
  // it, per WORKGROUP here (3000 of them) instead of once per kernel boundary; with an acquire per poll it was 2.54 ms.  The
  // chain stays the default; PXR_CHOL_ONE_LAUNCH=1 selects the single launch (tests/test_chol_gpu.py runs both).
  const bool step_launches = std::getenv("PXR_CHOL_ONE_LAUNCH") == nullptr || !one_launch_fits;
  if (int rc = hip_check(hipMemsetAsync(flags, 0, sizeof(int) * (step_launches ? (size_t)nblk : (size_t)nblk * (1 + kDoneStride)), st), "memset flags")) return rc;
  long long total_wgs = 0;
  for (int s = 0, k = 0; k < n; ++s, k += CNB) {
    const int nb = (n - k < CNB) ? n - k : CNB;
    const int rem = n_rows - k - nb;                      // rows below the diagonal tile (>= 1: the rhs row)
    const int n_panel = (CNB + rem + rows_per_wg - 1) / rows_per_wg;   // 64 identity rows + the rows below
    const int T = (n_rows - k + CNB - 1) / CNB;           // row tiles from block row s on; tiles (ti, tj), 1 <= tj <= ti < T
    const int n_update = s > 0 ? T * (T - 1) / 2 : 0;
    total_wgs += n_panel + n_update;
    if (step_launches)
      hipLaunchKernelGGL(k_chol_step, dim3(n_panel + n_update), dim3(512), kStepLds, st, a, n_rows, n, lda, s, n_panel, d_info, linv_ws,
                         linv_ws + (size_t)(nblk + 1) * CNB * CNB);
  }
  if (!step_launches) {
    if (total_wgs > 0x7fffffffll) return set_error(PXR_EINVAL, "chol_factor_solve: n = %d too large for one launch", n);
    if (int rc = hip_check(hipFuncSetAttribute((const void*)k_chol_all, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStepLds), "LDS size")) return rc;
    hipLaunchKernelGGL(k_chol_all, dim3((unsigned)total_wgs), dim3(512), kStepLds, st, a, n_rows, n, lda, d_info, linv_ws,
                       linv_ws + (size_t)(nblk + 1) * CNB * CNB, flags + nblk);
  }
  hipLaunchKernelGGL(k_chol_backsolve, dim3(nblk), dim3(256), 0, st, a, n, lda, linv_ws, x_out, flags, nblk);
  return hip_check(hipGetLastError(), "cholesky launch");
}

// pack a plain n x n (row-major, upper) system + rhs into the augmented layout and back
__global__ void k_aug_pack(int n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ aug) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int ld = n + 1;
  if (t >= (int64_t)(n + 1) * ld) return;
  const int r = (int)(t / ld), c = (int)(t % ld);
  double v = 0.0;
  if (r < n) v = (c == n) ? b[r] : (r <= c ? a[(size_t)r * n + c] : 0.0);
  aug[t] = v;
}
__global__ void k_aug_unpack(int n, const double* __restrict__ aug, const double* __restrict__ ldiag, double* __restrict__ a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n * n) return;
  const int r = (int)(t / n), c = (int)(t % n);
  // row-major upper (r, c) = column-major lower element (row c, column r); the diagonal tiles' factors live in the workspace
  if (r <= c) a[t] = (r / CNB == c / CNB) ? ldiag[(size_t)(r / CNB) * CNB * CNB + (c % CNB) + (r % CNB) * CNB] : aug[(size_t)r * (n + 1) + c];
}

}  // namespace pxr

// C-ABI: factor + solve.  d_a: n x n row-major with the UPPER triangle filled (the strictly lower part is
// ignored); on return the upper triangle holds the Cholesky factor (row-major upper = column-major lower L);
// d_b: right-hand side, overwritten by the solution.
extern "C" int pxr_dense_spd_solve(pxr_ctx* ctx, double* d_a, int n, double* d_b, int* h_info) {
  using namespace pxr;
  PXR_REQUIRE(ctx && d_a && d_b && h_info && n > 0, "pxr_dense_spd_solve: invalid argument");
  PXR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  double *aug = nullptr, *linv = nullptr;
  const size_t na = (size_t)(n + 1) * (n + 1), nl = chol_workspace_doubles(n);
  if (hipMalloc((void**)&aug, sizeof(double) * na) != hipSuccess || hipMalloc((void**)&linv, sizeof(double) * nl) != hipSuccess) {
    (void)hipFree(aug); (void)hipFree(linv);
    return set_error(PXR_ENOMEM, "pxr_dense_spd_solve: workspace allocation failed");
  }
  int* d_info = reinterpret_cast<int*>(ctx->d_scratch);
  hipLaunchKernelGGL(k_aug_pack, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, n, d_a, d_b, aug);
  int rc = chol_factor_solve(st, aug, n, d_info, linv, d_b, true);
  if (!rc) {
    hipLaunchKernelGGL(k_aug_unpack, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, st, n, aug,
                       linv + (size_t)((n + CNB - 1) / CNB + 1) * CNB * CNB, d_a);
    rc = hip_check(hipMemcpyAsync(h_info, d_info, sizeof(int), hipMemcpyDeviceToHost, st), "D2H info");
  }
  if (!rc) rc = hip_check(hipStreamSynchronize(st), "sync");
  (void)hipFree(aug); (void)hipFree(linv);
  return rc;
}

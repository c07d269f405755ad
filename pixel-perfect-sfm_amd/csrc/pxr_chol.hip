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
//   k_chol_panel   every workgroup factors the 64x64 diagonal block AND inverts its triangular
//                  factor in one sweep over column PAIRS on 4x4 register tiles (one barrier per pair, finished
//                  columns of L / rows of inv(L) exported to LDS so the updates need no masks);
//                  WG 0 writes L_kk and inv(L_kk) back, WG b >= 1 computes its 128 panel rows
//                  X = B inv(L_kk)^T as an LDS-tiled GEMM (8x4 register micro-tiles).
//   k_chol_syrk    trailing update A22 -= P P^T on 64x64 tiles (lower tiles only), fp64 MFMA.
//   k_chol_backsolve L^T x = y in one launch: a workgroup per block column, solution blocks handed on
//                  through flags; x_k = inv(L_kk)^T z_k, z_c -= L(k-block, c-block)^T x_k.
#include <hip/hip_runtime.h>

#include "pxr_internal.h"

namespace pxr {

constexpr int CNB = 64;
constexpr int PROWS = 128;   // panel rows per workgroup in the TRSM part

// n_rows = rows of the (augmented) matrix, n_cols = columns to factor, k = first column of the panel.
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ a, int n_rows, int n_cols, int lda, int k,
                                                    int* __restrict__ info, double* __restrict__ linv_out) {
  __shared__ double Lt[CNB][CNB + 2];     // Lt[m][j] = inv(L_kk)[j][m]; padded: a column write touches 16 rows, one bank group each
  __shared__ double Bs[CNB][PROWS];       // panel tile, Bs[m][row]
  __shared__ double cola[2][CNB], colc[2][CNB], rowa[2][CNB], rowc[2][CNB];
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  const int nb = min(CNB, n_cols - k);
#ifdef PXR_CHOL_PROFILE   // -DPXR_CHOL_PROFILE: workgroups 0 and 1 of the first panel print their phase times (100 MHz ticks)
  long long cp_t[6], cp_c[6]; int cp_n = 0;
  __shared__ long long cp_pair[33];
#define CHOL_PAIR() do { if (tid == 255) cp_pair[j >> 1] = clock64(); } while (0)   // thread 255: last wavefront, live to the end
#define CHOL_T() do { if (tid == 0) { cp_c[cp_n] = clock64(); cp_t[cp_n++] = wall_clock64(); } } while (0)
  CHOL_T();
#else
#define CHOL_T() do { } while (0)
#define CHOL_PAIR() do { } while (0)
#endif
  const bool is_panel_wg = blockIdx.x > 0;
  const int r0 = k + nb + ((int)blockIdx.x - 1) * PROWS;
  // issue the panel-tile loads first: they fly while the diagonal block is factored
  double breg[CNB * PROWS / 256];
  if (is_panel_wg) {
#pragma unroll
    for (int q = 0; q < CNB * PROWS / 256; ++q) {
      const int e = tid + q * 256, i = e % PROWS, m = e / PROWS;
      breg[q] = (r0 + i < n_rows && m < nb) ? a[(size_t)(r0 + i) + (size_t)(k + m) * lda] : 0.0;
    }
  }
  double Dt[4][4], Xt[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int R = 4 * ti + u, Cc = 4 * tj + w;
      double v = (R == Cc) ? 1.0 : 0.0;                     // identity padding beyond nb
      if (R < nb && Cc < nb && R >= Cc) v = a[(size_t)(k + R) + (size_t)(k + Cc) * lda];
      Dt[u][w] = v;
      Xt[u][w] = (R == Cc) ? 1.0 : 0.0;
    }
  int bad = 0;
  CHOL_T();
  // 32-step sweep over column PAIRS (j, j+1), one barrier per pair:
  //   L[:,j]   = D[:,j] / sqrt(D[j][j]);
  //   L[:,j+1] = (D[:,j+1] - L[:,j] L[j+1][j]) / sqrt(D[j+1][j+1] - L[j+1][j]^2);
  //   D       -= L[:,j] L[:,j]^T + L[:,j+1] L[:,j+1]^T;
  //   X[j]    /= L[j][j];  X[j+1] = (X[j+1] - L[j+1][j] X[j]) / L[j+1][j+1];  X -= L[:,j] X[j] + L[:,j+1] X[j+1]   (X: I -> inv(L))
#pragma unroll 2
  for (int j = 0; j < CNB; j += 2) {
    const int jb = j >> 2, jo = j & 3, buf = (j >> 1) & 1;   // jo is 0 or 2: both columns sit in the same 4-wide tile
    // a wavefront holds the tile rows 4 w .. 4 w + 3: once they are all finished (rows < j) it has nothing left to
    // export or update and only keeps the barrier company -- the sweep is bound by the LDS reads of the live ones
    const bool live = 4 * (tid >> 6) + 3 >= jb;
    CHOL_PAIR();
    if (live && tj == jb) {                                  // owners of columns j, j+1 of D
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        cola[buf][4 * ti + u] = jo == 0 ? Dt[u][0] : Dt[u][2];
        colc[buf][4 * ti + u] = jo == 0 ? Dt[u][1] : Dt[u][3];
      }
    }
    if (ti == jb) {                                          // owners of rows j, j+1 of X
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        rowa[buf][4 * tj + w] = jo == 0 ? Xt[0][w] : Xt[2][w];
        rowc[buf][4 * tj + w] = jo == 0 ? Xt[1][w] : Xt[3][w];
      }
    }
    __syncthreads();
    if (live) {
      // all LDS reads are issued unconditionally, back to back (one wait), and masked in registers:
      // conditional loads made the compiler emit a branch + wait per value
      const double d00 = cola[buf][j], d10 = cola[buf][j + 1], d11 = colc[buf][j + 1];
      double ua[4], wa[4], uc[4], wc[4], ra[4], rc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ua[u] = cola[buf][4 * ti + u]; wa[u] = cola[buf][4 * tj + u];
        uc[u] = colc[buf][4 * ti + u]; wc[u] = colc[buf][4 * tj + u];
        ra[u] = rowa[buf][4 * tj + u]; rc[u] = rowc[buf][4 * tj + u];
      }
      // the two pivots from INDEPENDENT reciprocal square roots:  1 / sqrt(d11 - d10^2 / d00) = sqrt(d00) / sqrt(d00 d11 - d10^2)
      const double det = fma(d00, d11, -(d10 * d10));
      if (!(d00 > 0.0) && bad == 0) bad = j + 1;
      if (!(det > 0.0) && bad == 0) bad = j + 2;
      const double inv0 = d00 > 0.0 ? rsqrt(d00) : 1.0;
      const double rdet = det > 0.0 ? rsqrt(det) : 1.0;
      const double l10 = d10 * inv0;
      const double inv1 = det > 0.0 ? rdet * (d00 * inv0) : 1.0;
      // No masks: once columns j, j+1 of L and rows j, j+1 of inv(L) have been exported to LDS (below), the
      // register entries of finished rows / columns are dead -- they are never read again, so the
      // rank-2 updates may overwrite them with garbage.
      double lia[4], lca[4], xra[4], lic[4], lcc[4], xrc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        lia[u] = ua[u] * inv0; lca[u] = wa[u] * inv0; xra[u] = ra[u] * inv0;
        lic[u] = fma(-lia[u], l10, uc[u]) * inv1;
        lcc[u] = fma(-lca[u], l10, wc[u]) * inv1;
        xrc[u] = fma(-l10, xra[u], rc[u]) * inv1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          Dt[u][w] = fma(-lic[u], lcc[w], fma(-lia[u], lca[w], Dt[u][w]));
          Xt[u][w] = fma(-lic[u], xrc[w], fma(-lia[u], xra[w], Xt[u][w]));
        }
      if (tj == jb && !is_panel_wg) {                        // columns j, j+1 of L (rows >= j / j+1 are meaningful)
#pragma unroll
        for (int u = 0; u < 4; ++u) { Bs[j][4 * ti + u] = lia[u]; Bs[j + 1][4 * ti + u] = lic[u]; }
      }
      if (ti == jb) {                                        // rows j, j+1 of inv(L): Lt[m][j] = inv(L)[j][m], zero for m > j
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          Lt[4 * tj + w][j] = (4 * tj + w <= j) ? xra[w] : 0.0;
          Lt[4 * tj + w][j + 1] = (4 * tj + w <= j + 1) ? xrc[w] : 0.0;
        }
      }
    }
  }
#ifdef PXR_CHOL_PROFILE
  if (tid == 255) cp_pair[32] = clock64();
#endif
  __syncthreads();
  CHOL_T();
  if (!is_panel_wg) {
    if (bad && bad <= nb && tid == 255) atomicCAS(info, 0, k + bad);   // the last wavefront is live (and tracks `bad`) to the end
    double* lo = linv_out + (size_t)(k / CNB) * CNB * CNB;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int R = 4 * ti + u, Cc = 4 * tj + w;
        if (R < nb && Cc < nb && R >= Cc) a[(size_t)(k + R) + (size_t)(k + Cc) * lda] = Bs[Cc][R];
        lo[R * CNB + Cc] = Lt[Cc][R];                        // inv(L_kk), row-major
      }
#ifdef PXR_CHOL_PROFILE
    CHOL_T();
    if (tid == 0 && k == 0) printf("chol panel wg0: loads %lld sweep %lld writeback %lld (x10 ns); sweep %lld shader cycles\n", cp_t[1] - cp_t[0], cp_t[2] - cp_t[1], cp_t[3] - cp_t[2], cp_c[2] - cp_c[1]);
    if (tid == 0 && k == 0) {
      printf("chol pairs (cycles):");
      for (int q = 0; q < 32; ++q) printf(" %lld", cp_pair[q + 1] - cp_pair[q]);
      printf("\n");
    }
#endif
    return;
  }
#pragma unroll
  for (int q = 0; q < CNB * PROWS / 256; ++q) {
    const int e = tid + q * 256;
    Bs[e / PROWS][e % PROWS] = breg[q];
  }
  __syncthreads();
  CHOL_T();
  // X = B inv(L)^T for this workgroup's PROWS rows: x[row][j] = sum_m Bs[m][row] Lt[m][j]
  const int tx = tid & 15, ty = tid >> 4;   // ty: 8 rows, tx: 4 columns
  double acc[8][4];
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[u][w] = 0.0;
#pragma unroll 4
  for (int m = 0; m < CNB; ++m) {
    double bv[8], lv[4];
#pragma unroll
    for (int u = 0; u < 8; ++u) bv[u] = Bs[m][8 * ty + u];
#pragma unroll
    for (int w = 0; w < 4; ++w) lv[w] = Lt[m][4 * tx + w];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) acc[u][w] = fma(bv[u], lv[w], acc[u][w]);
  }
  CHOL_T();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int j = 4 * tx + w;
    if (j >= nb) continue;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + 8 * ty + u;
      if (r < n_rows) a[(size_t)r + (size_t)(k + j) * lda] = acc[u][w];
    }
  }
#ifdef PXR_CHOL_PROFILE
  CHOL_T();
  if (tid == 0 && k == 0 && blockIdx.x == 1)
    printf("chol panel wg1: loads %lld sweep %lld stage %lld gemm %lld store %lld (x10 ns)\n", cp_t[1] - cp_t[0], cp_t[2] - cp_t[1], cp_t[3] - cp_t[2],
           cp_t[4] - cp_t[3], cp_t[5] - cp_t[4]);
#endif
}

// A22 -= P P^T on 64 x 64 tiles (lower tiles only); P = A(k+nb : n_rows, k : k+nb).  The one GEMM-shaped
// piece of the path: fp64 MFMA (v_mfma_f64_16x16x4_f64).  Each of the four wavefronts owns a 32 x 32
// quarter = 2 x 2 MFMA blocks, K = 64 in 16 steps.  Operands come from the LDS-staged panel rows
// (Pt[m][row]: lane l feeds row l % 16, k = l / 16 for A and for B alike).  The product is formed
// TRANSPOSED (A <- rows of the column tile, B <- rows of the row tile) so that the accumulator's
// lane index (col = lane & 15) runs along the matrix ROWS, which are contiguous in memory:
// C/D layout of the f64 MFMA is col = lane & 15, row = (lane >> 4) + 4 * reg.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
constexpr int SPAD = CNB + 8;   // LDS row stride: spreads the four k-rows of a step over the banks

__global__ __launch_bounds__(256) void k_chol_syrk(double* __restrict__ a, int n_rows, int lda, int k, int nb) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;
  __shared__ double Pi[CNB][SPAD];   // [m][row of the row tile]
  __shared__ double Pj[CNB][SPAD];   // [m][row of the column tile]
  const int r0 = k + nb;
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < CNB * CNB / 256; ++q) {
    const int e = tid + q * 256, i = e % CNB, m = e / CNB;
    const int gi = r0 + ti * CNB + i, gj = r0 + tj * CNB + i;
    // clamped address + select (a conditional load costs a branch and a wait per element)
    const size_t col = (size_t)(k + min(m, nb - 1)) * lda;
    const double vi = a[(size_t)min(gi, n_rows - 1) + col], vj = a[(size_t)min(gj, n_rows - 1) + col];
    Pi[m][i] = (gi < n_rows && m < nb) ? vi : 0.0;
    Pj[m][i] = (gj < n_rows && m < nb) ? vj : 0.0;
  }
  __syncthreads();
  const int lane = tid & 63, w = tid >> 6, wi = w >> 1, wj = w & 1;
  const int lr = lane & 15, lk = lane >> 4;
  mfma_d4 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) acc[x][y] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int s = 0; s < CNB / 4; ++s) {
    const int m = 4 * s + lk;
    const double aj0 = Pj[m][32 * wj + lr], aj1 = Pj[m][32 * wj + 16 + lr];       // A: column-tile rows
    const double bi0 = Pi[m][32 * wi + lr], bi1 = Pi[m][32 * wi + 16 + lr];       // B: row-tile rows
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj0, bi0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj0, bi1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj1, bi0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aj1, bi1, acc[1][1], 0, 0, 0);
  }
  // acc[x][y][r] = sum_m P[gj][m] P[gi][m]  with  gj = column-tile row 32 wj + 16 x + (lane >> 4) + 4 r,
  //                                               gi = row-tile row    32 wi + 16 y + (lane & 15)
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gj = r0 + tj * CNB + 32 * wj + 16 * x + lk + 4 * r;
        const int gi = r0 + ti * CNB + 32 * wi + 16 * y + lr;
        if (gi < n_rows && gj < n_rows && gi >= gj) a[(size_t)gi + (size_t)gj * lda] -= acc[x][y][r];
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

int chol_factor_solve(hipStream_t st, double* a, int n, int* d_info, double* linv_ws, double* x_out) {
  const int lda = n + 1, n_rows = n + 1;
  if (int rc = hip_check(hipMemsetAsync(d_info, 0, sizeof(int), st), "memset info")) return rc;
  for (int k = 0; k < n; k += CNB) {
    const int nb = (n - k < CNB) ? n - k : CNB;
    const int rem = n_rows - k - nb;   // rows below the diagonal block (>= 1: the rhs row)
    const int wgs = 1 + (rem + PROWS - 1) / PROWS;
    hipLaunchKernelGGL(k_chol_panel, dim3(wgs), dim3(256), 0, st, a, n_rows, n, lda, k, d_info, linv_ws);
    const int T = (rem + CNB - 1) / CNB;
    hipLaunchKernelGGL(k_chol_syrk, dim3(T, T), dim3(256), 0, st, a, n_rows, lda, k, nb);
  }
  const int nblk = (n + CNB - 1) / CNB;
  int* flags = reinterpret_cast<int*>(linv_ws + (size_t)nblk * CNB * CNB);   // spare block of the workspace
  if (int rc = hip_check(hipMemsetAsync(flags, 0, sizeof(int) * nblk, st), "memset flags")) return rc;
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
__global__ void k_aug_unpack(int n, const double* __restrict__ aug, double* __restrict__ a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n * n) return;
  const int r = (int)(t / n), c = (int)(t % n);
  if (r <= c) a[t] = aug[(size_t)r * (n + 1) + c];
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
  const size_t na = (size_t)(n + 1) * (n + 1), nl = (size_t)((n + 63) / 64 + 1) * 64 * 64;
  if (hipMalloc((void**)&aug, sizeof(double) * na) != hipSuccess || hipMalloc((void**)&linv, sizeof(double) * nl) != hipSuccess) {
    (void)hipFree(aug); (void)hipFree(linv);
    return set_error(PXR_ENOMEM, "pxr_dense_spd_solve: workspace allocation failed");
  }
  int* d_info = reinterpret_cast<int*>(ctx->d_scratch);
  hipLaunchKernelGGL(k_aug_pack, dim3((unsigned)((na + 255) / 256)), dim3(256), 0, st, n, d_a, d_b, aug);
  int rc = chol_factor_solve(st, aug, n, d_info, linv, d_b);
  if (!rc) {
    hipLaunchKernelGGL(k_aug_unpack, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, st, n, aug, d_a);
    rc = hip_check(hipMemcpyAsync(h_info, d_info, sizeof(int), hipMemcpyDeviceToHost, st), "D2H info");
  }
  if (!rc) rc = hip_check(hipStreamSynchronize(st), "sync");
  (void)hipFree(aug); (void)hipFree(linv);
  return rc;
}

// pxr_chol.hip -- dense fp64 Cholesky factorisation + solve of the reduced camera system on
// gfx950, written for the LM loop of pxr_ba_solve: no host synchronisation, no workspace
// allocation, a fixed short chain of launches (the library path -- rocSOLVER potrf/potrs --
// cost 24 ms + 8 ms of host-side overhead per call at n = 1593 for ~4 ms of kernels).
//
// Storage: the reduced system S is built row-major with only its UPPER triangle filled, which
// is the column-major LOWER triangle of the same buffer: A(i,j) = a[i + j*lda], i >= j.
// Right-looking blocked algorithm, NB = 64:
//   k_chol_panel  WG 0 factors the 64x64 diagonal block in LDS and writes it back; WG b >= 1
//                 factors it redundantly (cheaper than a launch boundary) and solves
//                 X L_kk^T = A(rows, k:k+NB) for its 128 rows as a small LDS-tiled GEMM with the
//                 explicit inverse of the triangular block (8x4 register micro-tiles).
//   k_chol_syrk   trailing update A22 -= P P^T on 64x64 tiles (lower tiles only), 4x4
//                 register micro-tiles, both panel tiles staged through LDS.
//   k_chol_solve  forward + backward substitution for one right-hand side by a single
//                 workgroup (the vector lives in LDS, the factor streams from L2).
#include <hip/hip_runtime.h>

#include "pxr_internal.h"

namespace pxr {

constexpr int CNB = 64;

// factor the NB x NB block held in LDS (lower), all 256 threads; two barriers per column.
// Pivots go to pv[] so that no thread overwrites D[j][j] while others still read it.
// Returns the first non-positive pivot (1-based) or 0.
__device__ int potf2_lds(double (*D)[CNB + 1], double* pv) {
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;
  int bad = 0;
  for (int j = 0; j < CNB; ++j) {
    const double djj = D[j][j];
    if (!(djj > 0.0) && bad == 0) bad = j + 1;
    const double piv = djj > 0.0 ? sqrt(djj) : 1.0;
    const double inv = 1.0 / piv;
    if (tid == j) pv[j] = piv;
    else if (tid > j && tid < CNB) D[tid][j] *= inv;
    __syncthreads();
    // trailing update D[i][c] -= D[i][j] * D[c][j], j < c <= i, 16 x 16 thread grid strided by 16
    for (int i = j + 1 + ti; i < CNB; i += 16) {
      const double lij = D[i][j];
      for (int c = j + 1 + tj; c <= i; c += 16) D[i][c] -= lij * D[c][j];
    }
    __syncthreads();
  }
  if (tid < CNB) D[tid][tid] = pv[tid];
  __syncthreads();
  return bad;
}

constexpr int PROWS = 128;   // panel rows per workgroup in the TRSM part

__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ a, int n, int lda, int k,
                                                    int* __restrict__ info) {
  __shared__ double D[CNB][CNB + 1];      // L_kk
  __shared__ double Lt[CNB][CNB];         // Lt[m][j] = inv(L_kk)[j][m]
  __shared__ double Bs[CNB][PROWS];       // panel tile, Bs[m][row]
  __shared__ double pv[CNB];
  const int tid = threadIdx.x;
  const int nb = min(CNB, n - k);
  for (int e = tid; e < CNB * CNB; e += blockDim.x) {   // diagonal block, identity padding beyond nb
    const int i = e % CNB, j = e / CNB;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < nb && j < nb && i >= j) v = a[(size_t)(k + i) + (size_t)(k + j) * lda];
    D[i][j] = v;
  }
  __syncthreads();
  const int bad = potf2_lds(D, pv);
  if (blockIdx.x == 0) {
    if (bad && tid == 0) atomicCAS(info, 0, k + bad);
    for (int e = tid; e < CNB * CNB; e += blockDim.x) {
      const int i = e % CNB, j = e / CNB;
      if (i < nb && j < nb && i >= j) a[(size_t)(k + i) + (size_t)(k + j) * lda] = D[i][j];
    }
    return;
  }
  // inverse of the triangular factor, row by row: Linv[i][c] = (d_ic - sum_{m=c}^{i-1} L[i][m] Linv[m][c]) / L[i][i]
  for (int e = tid; e < CNB * CNB; e += blockDim.x) Lt[e / CNB][e % CNB] = 0.0;
  __syncthreads();
  for (int i = 0; i < CNB; ++i) {
    if (tid <= i) {
      const int c = tid;
      double s = (c == i) ? 1.0 : 0.0;
      for (int m = c; m < i; ++m) s = fma(-D[i][m], Lt[c][m], s);   // Lt[c][m] = Linv[m][c]
      Lt[c][i] = s / D[i][i];
    }
    __syncthreads();
  }
  // X = B inv(L)^T for this workgroup's PROWS rows: x[row][j] = sum_m B[row][m] Linv[j][m] = sum_m Bs[m][row] Lt[m][j]
  const int r0 = k + CNB + (blockIdx.x - 1) * PROWS;
  for (int e = tid; e < CNB * PROWS; e += blockDim.x) {
    const int i = e % PROWS, m = e / PROWS;
    Bs[m][i] = (r0 + i < n && m < nb) ? a[(size_t)(r0 + i) + (size_t)(k + m) * lda] : 0.0;
  }
  __syncthreads();
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
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int j = 4 * tx + w;
    if (j >= nb) continue;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int r = r0 + 8 * ty + u;
      if (r < n) a[(size_t)r + (size_t)(k + j) * lda] = acc[u][w];
    }
  }
}

// A22 -= P P^T, tiles of 64 x 64; P = A(k+NB : n, k : k+NB)
__global__ __launch_bounds__(256) void k_chol_syrk(double* __restrict__ a, int n, int lda, int k) {
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;
  __shared__ double Pi[CNB][CNB];   // [m][row]
  __shared__ double Pj[CNB][CNB];
  const int r0 = k + CNB;
  const int tid = threadIdx.x;
  const int nb = CNB;               // a full panel precedes every non-empty trailing block
  for (int e = tid; e < CNB * CNB; e += 256) {
    const int i = e % CNB, m = e / CNB;
    const int gi = r0 + ti * CNB + i, gj = r0 + tj * CNB + i;
    Pi[m][i] = (gi < n) ? a[(size_t)gi + (size_t)(k + m) * lda] : 0.0;
    Pj[m][i] = (gj < n) ? a[(size_t)gj + (size_t)(k + m) * lda] : 0.0;
  }
  __syncthreads();
  const int tx = tid & 15, ty = tid >> 4;
  double acc[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 8
  for (int m = 0; m < nb; ++m) {
    double pi[4], pj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { pi[u] = Pi[m][4 * ty + u]; pj[u] = Pj[m][4 * tx + u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) acc[u][w] = fma(pi[u], pj[w], acc[u][w]);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int gj = r0 + tj * CNB + 4 * tx + w;
    if (gj >= n) continue;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int gi = r0 + ti * CNB + 4 * ty + u;
      if (gi < n && gi >= gj) a[(size_t)gi + (size_t)gj * lda] -= acc[u][w];
    }
  }
}

// single right-hand side: L y = b, L^T x = y; one workgroup of 1024 threads, b in LDS
__global__ __launch_bounds__(1024) void k_chol_solve(const double* __restrict__ a, int n, int lda,
                                                     double* __restrict__ b) {
  extern __shared__ double xs[];   // n doubles
  __shared__ double Dd[CNB][CNB + 1];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < n; i += nt) xs[i] = b[i];
  __syncthreads();
  // forward
  for (int k = 0; k < n; k += CNB) {
    const int nb = min(CNB, n - k);
    for (int e = tid; e < CNB * CNB; e += nt) {
      const int i = e % CNB, j = e / CNB;
      Dd[i][j] = (i < nb && j < nb && i >= j) ? a[(size_t)(k + i) + (size_t)(k + j) * lda] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {   // one wave: sequential over columns, lanes = rows (wave-synchronous)
      double v = (tid < nb) ? xs[k + tid] : 0.0;
      for (int j = 0; j < nb; ++j) {
        const double yj = __shfl(v, j) / Dd[j][j];
        if (tid == j) v = yj;
        else if (tid > j) v = fma(-Dd[tid][j], yj, v);
      }
      if (tid < nb) xs[k + tid] = v;
    }
    __syncthreads();
    // xs[i] -= sum_j A(i, k+j) y_j for i >= k + nb : one thread per row, coalesced over rows
    for (int i = k + nb + tid; i < n; i += nt) {
      double s = xs[i];
      for (int j = 0; j < nb; ++j) s = fma(-a[(size_t)i + (size_t)(k + j) * lda], xs[k + j], s);
      xs[i] = s;
    }
    __syncthreads();
  }
  // backward: L^T x = y
  const int nblk = (n + CNB - 1) / CNB;
  for (int bk = nblk - 1; bk >= 0; --bk) {
    const int k = bk * CNB, nb = min(CNB, n - k);
    // xs[k+j] -= sum_{i >= k+nb} A(i, k+j) x_i : a wave per column (columns are contiguous)
    {
      const int lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
      for (int j = wv; j < nb; j += nw) {
        double s = 0.0;
        for (int i = k + nb + lane; i < n; i += 64) s = fma(a[(size_t)i + (size_t)(k + j) * lda], xs[i], s);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) xs[k + j] -= s;
      }
      __syncthreads();
    }
    for (int e = tid; e < CNB * CNB; e += nt) {
      const int i = e % CNB, j = e / CNB;
      Dd[i][j] = (i < nb && j < nb && i >= j) ? a[(size_t)(k + i) + (size_t)(k + j) * lda] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    if (tid < 64) {
      double v = (tid < nb) ? xs[k + tid] : 0.0;
      for (int j = nb - 1; j >= 0; --j) {
        const double xj = __shfl(v, j) / Dd[j][j];
        if (tid == j) v = xj;
        else if (tid < j) v = fma(-Dd[j][tid], xj, v);   // L^T(tid, j) = L(j, tid)
      }
      if (tid < nb) xs[k + tid] = v;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) b[i] = xs[i];
}

// Enqueue the factorisation of the n x n SPD matrix (column-major lower / row-major upper).
// *d_info (device int, zeroed here) receives the 1-based index of the first non-positive pivot.
int chol_factor(hipStream_t st, double* a, int n, int lda, int* d_info) {
  if (int rc = hip_check(hipMemsetAsync(d_info, 0, sizeof(int), st), "memset info")) return rc;
  for (int k = 0; k < n; k += CNB) {
    const int rem = n - k - CNB;   // rows below the diagonal block
    const int wgs = 1 + (rem > 0 ? (rem + PROWS - 1) / PROWS : 0);
    hipLaunchKernelGGL(k_chol_panel, dim3(wgs), dim3(256), 0, st, a, n, lda, k, d_info);
    if (rem > 0) {
      const int T = (rem + CNB - 1) / CNB;
      hipLaunchKernelGGL(k_chol_syrk, dim3(T, T), dim3(256), 0, st, a, n, lda, k);
    }
  }
  return hip_check(hipGetLastError(), "cholesky launch");
}

int chol_solve(hipStream_t st, const double* a, int n, int lda, double* b) {
  const size_t shmem = sizeof(double) * (size_t)n;
  if (shmem > 120 * 1024) return set_error(PXR_EUNSUPPORTED, "chol_solve: n = %d exceeds the LDS-resident limit", n);
  if (shmem > 48 * 1024) {
    if (int rc = hip_check(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chol_solve),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                           "hipFuncSetAttribute(k_chol_solve)"))
      return rc;
  }
  hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(1024), shmem, st, a, n, lda, b);
  return hip_check(hipGetLastError(), "cholesky solve launch");
}

}  // namespace pxr

// C-ABI: factor + solve in place.  d_a: n x n row-major with the UPPER triangle filled (the strictly
// lower part is ignored and overwritten); d_b: right-hand side, overwritten by the solution.
extern "C" int pxr_dense_spd_solve(pxr_ctx* ctx, double* d_a, int n, double* d_b, int* h_info) {
  PXR_REQUIRE(ctx && d_a && d_b && h_info && n > 0, "pxr_dense_spd_solve: invalid argument");
  PXR_HIP(hipSetDevice(ctx->device));
  int* d_info = reinterpret_cast<int*>(ctx->d_scratch);
  if (int rc = pxr::chol_factor(ctx->stream, d_a, n, n, d_info)) return rc;
  if (int rc = pxr::chol_solve(ctx->stream, d_a, n, n, d_b)) return rc;
  PXR_HIP(hipMemcpyAsync(h_info, d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PXR_HIP(hipStreamSynchronize(ctx->stream));
  return PXR_OK;
}

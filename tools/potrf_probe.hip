// potrf_probe.hip -- standalone bench + check of the tall-panel factorisation core of csrc/pxr_chol.hip
// (pxr_chol_core.h): one workgroup of 8 wavefronts factors a 64 x 64 SPD tile and carries 64 extra rows
// through the same sweep (here: identity rows, which come out as inv(L)^T).  Prints shader cycles and the error
// against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Ipixel-perfect-sfm_amd/csrc -o tools/bin/potrf_probe tools/potrf_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pxr_chol_core.h"   // build with -DPXR_CHOL_PROFILE for the per-phase stamps

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using namespace pxr::cholcore;

__global__ __launch_bounds__(512) void k_probe(const double* __restrict__ a, int lda, int nb, double* __restrict__ l_out,
                                               double* __restrict__ linv_out, long long* cyc, int* info, long long* prof) {
  __shared__ FactorLds lds;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  d4 acc[4];
  RowSink sink;
  sink.l_diag = blockIdx.x == 0 ? l_out : nullptr; sink.ld = 64; sink.nb = nb; sink.rows_valid = 16; sink.rows_out = nullptr;
  sink.mode = RowSink::kNone;
#ifdef PXR_CHOL_PROFILE
  sink.prof = prof;
#endif
  if (wave < 4) load_diag_blocks(acc, a, lda, nb, wave, lane);
  else if (wave >= ROW_WAVE0) {
    const int vb = ROW_WAVES * blockIdx.x + (wave - ROW_WAVE0);     // 16-row block of the identity this wavefront carries
    if (vb < 4) { identity_rows(acc, vb, lane); sink.mode = RowSink::kInverse; sink.rows_out = linv_out + 16 * vb; }
  }
  __syncthreads();
  const long long t0 = clock64();
  int bad = factor_tile(lds, acc, wave, lane, sink);
  const long long t1 = clock64();
  if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; info[0] = bad; cyc[1] = t0; }
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 64;
  const int lda = 64;
  std::vector<double> A(64 * 64, 0.0), M(64 * 70);
  srand(3);
  for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
  for (int i = 0; i < nb; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < 70; ++k) s += M[i * 70 + k] * M[j * 70 + k];
      A[i + j * lda] = s;   // column-major lower
    }
  // host Cholesky + inverse
  std::vector<double> L(64 * 64, 0.0), Li(64 * 64, 0.0);
  for (int j = 0; j < nb; ++j) {
    double d = A[j + j * lda];
    for (int k = 0; k < j; ++k) d -= L[j * 64 + k] * L[j * 64 + k];
    L[j * 64 + j] = std::sqrt(d);
    for (int i = j + 1; i < nb; ++i) {
      double s = A[i + j * lda];
      for (int k = 0; k < j; ++k) s -= L[i * 64 + k] * L[j * 64 + k];
      L[i * 64 + j] = s / L[j * 64 + j];
    }
  }
  for (int j = nb; j < 64; ++j) L[j * 64 + j] = 1.0;
  for (int c = 0; c < 64; ++c) {          // L Li = I, column by column
    for (int r = c; r < 64; ++r) {
      double s = (r == c) ? 1.0 : 0.0;
      for (int k = c; k < r; ++k) s -= L[r * 64 + k] * Li[k * 64 + c];
      Li[r * 64 + c] = s / L[r * 64 + r];
    }
  }
  double *dA, *dL, *dLi; long long* dc; int* di; long long* dp; CHK(hipMalloc(&dp, 8 * 8 * 8 * 4));
  CHK(hipMalloc(&dA, 8 * 64 * 64)); CHK(hipMalloc(&dL, 8 * 64 * 64)); CHK(hipMalloc(&dLi, 8 * 64 * 64)); CHK(hipMalloc(&dc, 64)); CHK(hipMalloc(&di, 64));
  CHK(hipMemcpy(dA, A.data(), 8 * 64 * 64, hipMemcpyHostToDevice));
  CHK(hipMemset(dL, 0, 8 * 64 * 64)); CHK(hipMemset(dLi, 0, 8 * 64 * 64));
  long long cyc = 0; int info = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_probe, dim3(2), dim3(512), 0, 0, dA, lda, nb, dL, dLi, dc, di, dp);
    CHK(hipDeviceSynchronize());
  }
  CHK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(&info, di, 4, hipMemcpyDeviceToHost));
  std::vector<double> gL(64 * 64), gLi(64 * 64);
  CHK(hipMemcpy(gL.data(), dL, 8 * 64 * 64, hipMemcpyDeviceToHost)); CHK(hipMemcpy(gLi.data(), dLi, 8 * 64 * 64, hipMemcpyDeviceToHost));
  double eL = 0, eI = 0, sL = 0, sI = 0;
  for (int r = 0; r < nb; ++r)
    for (int c = 0; c <= r; ++c) {
      eL = std::fmax(eL, std::fabs(gL[r + c * 64] - L[r * 64 + c])); sL = std::fmax(sL, std::fabs(L[r * 64 + c]));
    }
  for (int r = 0; r < 64; ++r)
    for (int c = 0; c < 64; ++c) {
      eI = std::fmax(eI, std::fabs(gLi[r * 64 + c] - Li[r * 64 + c])); sI = std::fmax(sI, std::fabs(Li[r * 64 + c]));
    }
#ifdef PXR_CHOL_PROFILE
  {
    long long hp[8 * 8 * 4], t0[2];
    CHK(hipMemcpy(hp, dp, sizeof(hp), hipMemcpyDeviceToHost)); CHK(hipMemcpy(t0, dc, 16, hipMemcpyDeviceToHost));
    const int waves[] = {0, 1, 3, 5, 7};
    for (int w : waves) {
      printf("wave %d (phase-1 work | wait A | phase-2 work | wait B):", w);
      long long prev = t0[1];
      for (int p = 0; p < 8; ++p) {
        const long long* q = hp + (w * 8 + p) * 4;
        printf("  %lld|%lld|%lld|%lld", q[0] - prev, q[1] - q[0], q[2] - q[1], q[3] - q[2]);
        prev = q[3];
      }
      printf("\n");
    }
  }
#endif
  if (getenv("PROBE_DUMP")) {
    for (int r = 0; r < 20; ++r) { printf("row %2d got:", r); for (int c = 0; c < 12; ++c) printf(" % .4f", gLi[r * 64 + c]); printf("\n       want:"); for (int c = 0; c < 12; ++c) printf(" % .4f", Li[r * 64 + c]); printf("\n"); }
  }
  printf("nb %d: factor_tile %lld shader cycles (%.1f per pivot), info %d, max |dL| %.3e (scale %.2e), max |dLinv| %.3e (scale %.2e)\n", nb, cyc,
         cyc / 64.0, info, eL, sL, eI, sI);
  return (eL < 1e-11 * sL && eI < 1e-9 * sI && info == 0) ? 0 : 1;
}

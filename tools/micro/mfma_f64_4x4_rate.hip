// Microbenchmark: issue rate of v_mfma_f64_4x4x4_4b_f64 (four 4 x 4 x 4 blocks, 512 flop) next to v_mfma_f64_16x16x4_f64 (2 048 flop).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, bool SMALL>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  double accs[NACC]; d4 accb[NACC];
  for (int j = 0; j < NACC; ++j) { accs[j] = 0.0; accb[j] = d4{0, 0, 0, 0}; }
  double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      if (SMALL) accs[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, accs[j], 0, 0, 0);
      else accb[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, accb[j], 0, 0, 0);
    }
  }
  double s = 0;
  for (int j = 0; j < NACC; ++j) s += accs[j] + accb[j][0] + accb[j][1] + accb[j][2] + accb[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool SMALL>
void run(int wps, double* d_out) {
  const int iters = 4000, blocks = 256 * wps;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(256), 0, 0, d_out, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, SMALL>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * NACC * wps;
  const double flops = mf * (SMALL ? 512.0 : 2048.0) * 1024.0;
  printf("%s acc %d waves/SIMD %d : %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD\n", SMALL ? "4x4x4_4b " : "16x16x4  ", NACC, wps, ms,
         flops / (ms * 1e-3) / 1e12, ms * 1e6 / mf);
}
int main() {
  double* d_out;
  (void)hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8);
  for (int wps : {2, 4}) { run<4, true>(wps, d_out); run<8, true>(wps, d_out); run<4, false>(wps, d_out); }
  return 0;
}

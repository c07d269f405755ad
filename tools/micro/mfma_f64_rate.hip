// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- NACC independent accumulators per wavefront, WPS wavefronts
// per SIMD; prints shader cycles per MFMA per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, long long* cyc) {
  d4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = d4{0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = blockIdx.x * 1e-3 + 1.0;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[j], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int wps, double* d_out, long long* d_cyc) {
  const int iters = 4000;
  const int blocks = 256 * wps;   // 256-thread workgroups = 4 wavefronts = one per SIMD; wps workgroups per CU
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d_out, 10, d_cyc);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_cyc);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long cyc; (void)hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
  const double mf = (double)iters * NACC * wps;     // MFMAs per SIMD
  const double flops = mf * 2048.0 * 1024.0;
  printf("acc %d  waves/SIMD %d : %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD  (s_memtime ticks of wave 0: %lld = %.2f per own MFMA)\n",
         NACC, wps, ms, flops / (ms * 1e-3) / 1e12, ms * 1e6 / mf, cyc, (double)cyc / (iters * NACC));
}
int main() {
  double* d_out; long long* d_cyc;
  (void)hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8); (void)hipMalloc(&d_cyc, 8);
  for (int wps : {1, 2, 4}) { run<1>(wps, d_out, d_cyc); run<2>(wps, d_out, d_cyc); run<4>(wps, d_out, d_cyc); }
  return 0;
}

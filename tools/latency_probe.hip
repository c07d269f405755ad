// latency_probe.hip -- the latencies a SERIAL chain pays on the GPU it runs on (one workgroup, shader cycles per step):
//   dependent v_fma_f64 / v_mul_f64 / v_rsq_f64 chains of ONE wavefront, the rsqrt() of the device library,
//   an LDS write -> s_barrier -> LDS read round trip of a workgroup of 4 and of 8 wavefronts, s_barrier alone.
// The Cholesky panel sweep (csrc/pxr_chol.hip) is such a chain; these numbers are its floor.
//   hipcc --offload-arch=gfx950 -O3 -o latency_probe tools/latency_probe.hip && ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int N = 2048;

__global__ void k_fma(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x, b = seed * 0.5;
  const long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x) : "v"(b));
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (x == 1234.5) out[0] = x;
}
__global__ void k_mul(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x, b = 1.0000001;
  const long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (x == 1234.5) out[0] = x;
}
__global__ void k_rsq(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x;
  const long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) asm volatile("v_rsq_f64 %0, %0\n s_nop 0" : "+v"(x));
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (x == 1234.5) out[0] = x;
}
__global__ void k_rsqrt_lib(double* out, long long* cyc, double seed) {
  double x = seed + threadIdx.x;
  const long long t0 = clock64();
  for (int i = 0; i < N; ++i) x = rsqrt(x) + 1.0;
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (x == 1234.5) out[0] = x;
}
__global__ void k_cndmask(double* out, long long* cyc, double seed) {
  float x = (float)seed + threadIdx.x, b = 3.0f;
  const long long t0 = clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (x == 1234.5f) out[0] = x;
}
// every step: each thread writes one double, barrier, reads its neighbour's
__global__ void k_lds_round(double* out, long long* cyc, double seed) {
  __shared__ double sh[2][512];
  double x = seed + threadIdx.x;
  const int nxt = (threadIdx.x + 17) % blockDim.x;
  const long long t0 = clock64();
  for (int i = 0; i < N; ++i) {
    sh[i & 1][threadIdx.x] = x;
    __syncthreads();
    x = sh[i & 1][nxt] + 1.0;
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (x == 1234.5) out[0] = x;
}
// the same with the reads of the Cholesky pair step: 5 x 16-byte reads behind the barrier
__global__ void k_lds_round_b128(double* out, long long* cyc, double seed) {
  __shared__ double sh[2][1024];
  double x = seed + threadIdx.x;
  const int t = threadIdx.x & 255, a0 = 4 * (t >> 4), a1 = 4 * (t & 15);
  const long long t0 = clock64();
  for (int i = 0; i < N; ++i) {
    if ((t & 15) == (i & 15)) { sh[i & 1][a0] = x; sh[i & 1][a0 + 1] = x; sh[i & 1][256 + a0] = x; sh[i & 1][256 + a0 + 1] = x; }
    __syncthreads();
    const double* s = sh[i & 1];
    x += s[a0] + s[a0 + 1] + s[a0 + 2] + s[a0 + 3] + s[a1] + s[a1 + 1] + s[a1 + 2] + s[a1 + 3] + s[256 + a0] + s[256 + a0 + 1] + s[256 + a0 + 2] +
         s[256 + a0 + 3] + s[256 + a1] + s[256 + a1 + 1] + s[256 + a1 + 2] + s[256 + a1 + 3] + s[i & 63];
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  if (x == 1234.5) out[0] = x;
}
__global__ void k_barrier(double* out, long long* cyc, double seed) {
  const long long t0 = clock64();
  for (int i = 0; i < N; ++i) __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double* out; long long* cyc; long long h;
  CHK(hipMalloc(&out, 64)); CHK(hipMalloc(&cyc, 64));
#define RUN(K, THREADS, LABEL) do { for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(K, dim3(1), dim3(THREADS), 0, 0, out, cyc, 1.5); CHK(hipDeviceSynchronize()); } \
    CHK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); printf("%-58s %7.1f cycles per step\n", LABEL, (double)h / N); } while (0)
  RUN(k_fma, 64, "dependent v_fma_f64, one wavefront");
  RUN(k_mul, 64, "dependent v_mul_f64, one wavefront");
  RUN(k_rsq, 64, "dependent v_rsq_f64, one wavefront");
  RUN(k_rsqrt_lib, 64, "dependent rsqrt(double) + add, one wavefront");
  RUN(k_cndmask, 64, "dependent v_add_f32, one wavefront");
  RUN(k_barrier, 256, "s_barrier, 4 wavefronts");
  RUN(k_barrier, 512, "s_barrier, 8 wavefronts");
  RUN(k_lds_round, 256, "LDS write -> barrier -> read (8 B), 4 wavefronts");
  RUN(k_lds_round, 512, "LDS write -> barrier -> read (8 B), 8 wavefronts");
  RUN(k_lds_round_b128, 256, "LDS owners write -> barrier -> 17 reads, 4 wavefronts");
  RUN(k_lds_round_b128, 512, "LDS owners write -> barrier -> 17 reads, 8 wavefronts");
  return 0;
}

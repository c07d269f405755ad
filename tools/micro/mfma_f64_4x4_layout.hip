// Probe: operand / result layout of v_mfma_f64_4x4x4_4b_f64 (four blocks of 4 x 4 x 4) on gfx950: A = unit vector at lane p,
// B = unit vector at lane q -> which lane of D receives the product?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int l = threadIdx.x;
  for (int p = 0; p < 64; ++p)
    for (int q = 0; q < 64; ++q) {
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == p ? 1.0 : 0.0, l == q ? 1.0 : 0.0, 0.0, 0, 0, 0);
      if (d != 0.0) out[p * 64 + q] = l;
    }
}
int main() {
  int h[4096], *d;
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMemset(d, 0xff, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int p = 0; p < 64; ++p) {
    printf("A lane %2d meets B lanes:", p);
    for (int q = 0; q < 64; ++q) if (h[p * 64 + q] >= 0) printf(" %d->D%d", q, h[p * 64 + q]);
    printf("\n");
  }
  return 0;
}

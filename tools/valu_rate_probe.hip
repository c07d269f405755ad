// valu_rate_probe.hip -- issue rate of the vector instructions the fp64 kernels are made of, on the GPU it runs on.
// Each kernel runs 8 independent dependency chains of ONE instruction per lane, 4096 x 8 times, with 8 wavefronts per SIMD
// resident, and reports SIMD cycles per wavefront instruction relative to v_fma_f32 (4 cycles: 64 lanes over 16 ALUs).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate_probe tools/valu_rate_probe.hip && ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITER = 4096;

#define PROBE(NAME, TYPE, INIT, ASM)                                                             \
  __global__ __launch_bounds__(256) void NAME(TYPE* out, TYPE seed) {                            \
    TYPE r[8];                                                                                   \
    for (int q = 0; q < 8; ++q) r[q] = INIT;                                                     \
    TYPE b = seed;                                                                               \
    for (int it = 0; it < ITER; ++it) {                                                          \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) asm volatile(ASM : "+v"(r[q]) : "v"(b));     \
    }                                                                                            \
    TYPE s = r[0];                                                                               \
    for (int q = 1; q < 8; ++q) s += r[q];                                                       \
    if (s == (TYPE)12345.678) out[0] = s;                                                        \
  }

PROBE(p_fma_f32, float, (float)(threadIdx.x + q), "v_fma_f32 %0, %0, %1, %0")
PROBE(p_pk_fma_f32, double, (double)(threadIdx.x + q), "v_pk_fma_f32 %0, %0, %1, %0")
PROBE(p_fma_f64, double, (double)(threadIdx.x + q), "v_fma_f64 %0, %0, %1, %0")
PROBE(p_add_f64, double, (double)(threadIdx.x + q), "v_add_f64 %0, %0, %1")
PROBE(p_mul_f64, double, (double)(threadIdx.x + q), "v_mul_f64 %0, %0, %1")
PROBE(p_pk_add_f16, float, (float)(threadIdx.x + q), "v_pk_add_f16 %0, %0, %1")
PROBE(p_mov_dpp, float, (float)(threadIdx.x + q), "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
PROBE(p_cvt_f32_f16, float, (float)(threadIdx.x + q), "v_cvt_f32_f16 %0, %1")
PROBE(p_fma_mix, float, (float)(threadIdx.x + q), "v_fma_mix_f32 %0, %1, %1, %0 op_sel_hi:[1,1,0]")
PROBE(p_dot2_f32_f16, float, (float)(threadIdx.x + q), "v_dot2_f32_f16 %0, %1, %1, %0")


// three DISTINCT source registers per instruction (the register-file ports, not only the ALU, set the rate)
#define PROBE3(NAME, TYPE, ASM)                                                                   \
  __global__ __launch_bounds__(256) void NAME(TYPE* out, TYPE seed) {                            \
    TYPE r[8], x[8], y[8];                                                                       \
    for (int q = 0; q < 8; ++q) { r[q] = (TYPE)(threadIdx.x + q); x[q] = seed + (TYPE)q; y[q] = seed * (TYPE)(q + 2); } \
    for (int it = 0; it < ITER; ++it) {                                                          \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) asm volatile(ASM : "+v"(r[q]) : "v"(x[q]), "v"(y[(q + 3) & 7])); \
    }                                                                                            \
    TYPE s = r[0];                                                                               \
    for (int q = 1; q < 8; ++q) s += r[q];                                                       \
    if (s == (TYPE)12345.678) out[0] = s;                                                        \
  }
PROBE3(p3_fmac_f64, double, "v_fmac_f64 %0, %1, %2")
PROBE3(p3_fma_f64, double, "v_fma_f64 %0, %1, %2, %0")
PROBE3(p3_add_f64, double, "v_add_f64 %0, %1, %2")
PROBE3(p3_mul_f64, double, "v_mul_f64 %0, %1, %2")
PROBE3(p3_fma_f32, float, "v_fma_f32 %0, %1, %2, %0")
PROBE3(p3_fmac_f32, float, "v_fmac_f32 %0, %1, %2")
PROBE3(p3_pk_add_f16, float, "v_pk_add_f16 %0, %1, %2")
PROBE3(p3_cvt_sdwa, float, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")

// conversions between widths need differently typed operands
__global__ __launch_bounds__(256) void p_cvt_f64_f32(double* out, float seed) {
  double r[8]; float b = seed + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[q]) : "v"(b));
  }
  double s = 0; for (int q = 0; q < 8; ++q) s += r[q];
  if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void p_cvt_f32_f64(float* out, double seed) {
  float r[8]; double b = seed + threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[q]) : "v"(b));
  }
  float s = 0; for (int q = 0; q < 8; ++q) s += r[q];
  if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void p_permlane32_swap(float* out, float seed) {
  unsigned r[8];
  for (int q = 0; q < 8; ++q) r[q] = threadIdx.x + q;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r[q]), "+v"(r[q + 1]));
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r[q]), "+v"(r[q + 1]));
    }
  }
  unsigned s = 0; for (int q = 0; q < 8; ++q) s += r[q];
  if (s == 0x12345678u) out[0] = (float)s;
}

static bool g_long = false;
template <typename K, typename... A>
static int run(const char* name, K kern, double base_ns, double* ns_out, A... args) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
  const int wgs = prop.multiProcessorCount * 8;        // 8 workgroups x 4 wavefronts per CU = 8 wavefronts per SIMD
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, args...);
  CHK(hipDeviceSynchronize());
  float best = 1e30f;
  const int reps = g_long ? 1500 : 3;                  // --long: ~0.7 s per instruction, for sampling the clock beside it
  if (g_long) { printf("[start %s]\n", name); fflush(stdout); }
  for (int rep = 0; rep < reps; ++rep) {
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, args...);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double per = best * 1e6 / ((double)ITER * 8 * 8);   // ns per wavefront instruction per SIMD (8 waves share a SIMD)
  *ns_out = per;
  if (base_ns > 0) printf("%-22s %8.3f ms   %6.2f cycles per wavefront instruction (v_fma_f32 = 4)\n", name, best, 4.0 * per / base_ns);
  else printf("%-22s %8.3f ms   %6.3f ns per wavefront instruction per SIMD -> %.2f GHz if 4 cycles\n", name, best, per, 4.0 / per);
  return 0;
}

int main(int argc, char** argv) {
  g_long = argc > 1;
  void* out; CHK(hipMalloc(&out, 64));
  double base = 0, t;
  if (run("v_fma_f32", p_fma_f32, 0.0, &base, (float*)out, 1.0f)) return 1;
  run("v_pk_fma_f32", p_pk_fma_f32, base, &t, (double*)out, 1.0);
  run("v_fma_f64", p_fma_f64, base, &t, (double*)out, 1.0);
  run("v_add_f64", p_add_f64, base, &t, (double*)out, 1.0);
  run("v_mul_f64", p_mul_f64, base, &t, (double*)out, 1.0);
  run("v_cvt_f64_f32", p_cvt_f64_f32, base, &t, (double*)out, 1.0f);
  run("v_cvt_f32_f64", p_cvt_f32_f64, base, &t, (float*)out, 1.0);
  run("v_cvt_f32_f16", p_cvt_f32_f16, base, &t, (float*)out, 1.0f);
  run("v_pk_add_f16", p_pk_add_f16, base, &t, (float*)out, 1.0f);
  run("v_fma_mix_f32", p_fma_mix, base, &t, (float*)out, 1.0f);
  run("v_dot2_f32_f16", p_dot2_f32_f16, base, &t, (float*)out, 1.0f);
  run("v_mov_b32_dpp", p_mov_dpp, base, &t, (float*)out, 1.0f);
  run("v_fmac_f64 3 regs", p3_fmac_f64, base, &t, (double*)out, 1.0);
  run("v_fma_f64 3 regs", p3_fma_f64, base, &t, (double*)out, 1.0);
  run("v_add_f64 3 regs", p3_add_f64, base, &t, (double*)out, 1.0);
  run("v_mul_f64 3 regs", p3_mul_f64, base, &t, (double*)out, 1.0);
  run("v_fma_f32 3 regs", p3_fma_f32, base, &t, (float*)out, 1.0f);
  run("v_fmac_f32 3 regs", p3_fmac_f32, base, &t, (float*)out, 1.0f);
  run("v_pk_add_f16 3 regs", p3_pk_add_f16, base, &t, (float*)out, 1.0f);
  run("v_cvt_f32_f16_sdwa", p3_cvt_sdwa, base, &t, (float*)out, 1.0f);
  run("v_permlane32_swap(+nop)", p_permlane32_swap, base, &t, (float*)out, 1.0f);
  return 0;
}

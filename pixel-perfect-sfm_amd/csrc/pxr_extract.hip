// pxr_extract.hip -- sparse patch producer: dense feature map (device, CHW) -> arena patches (HWC).
//
// Reference path: FeatureExtractor.tensor_to_fmap, sparse branch
// (pixsfm/features/extractor.py:152-199): L2-normalise the map over channels, cast to the storage
// dtype, corners = clip(int(kp * scale - ps / 2), 0, (w, h) - ps - 1), gather ps x ps windows with
// extract_patches_torch and permute to [n][ps][ps][C] -- then copy GPU -> CPU numpy, which the
// reference itself flags as "main performance bottleneck" (features/extract_patches.py:41-44), and
// later back to wherever the optimiser runs.  Here the gather writes straight into the HBM arena the
// KA / BA kernels read; nothing crosses PCIe.
//
// Patch ORDER (round 2): keypoints arrive in detector order, i.e. scattered over the map.  A 128 x 256 x 320 fp32 map
// is 42 MB -- far beyond the 4 MB L2 of an XCD -- and every map texel is gathered ~300 times by overlapping patches, so
// in keypoint order nearly every 64-byte row segment came from the Infinity Cache / HBM again (SQ counters: 5 % VALU,
// 93 % of the wave time parked on memory).  The workgroups therefore walk the patches in MAP-TILE order (a counting
// sort of the corners by 16 x 16 tile, three tiny kernels): neighbouring workgroups read neighbouring rows, the L2
// serves the reuse, and the output slot of a patch is still its keypoint index.
//
// HBM-bound transposition: one workgroup per patch, one patch row per step.  Reads are the CHW
// rows (16 contiguous x per channel = 64 B at fp32), staged through a padded LDS tile so that the
// writes are the arena's channel-fastest texels (256 B per pixel at fp16, 16 B per lane).
#include <hip/hip_runtime.h>

#include "pxr_device.h"
#include "pxr_internal.h"

namespace pxr {

// corner of keypoint k in map texels (extractor.py:192-193): C-style truncation like astype(np.int32)
__device__ __forceinline__ void ex_corner(const double* kp, double sx, double sy, int ps, int w, int h, int& x0, int& y0) {
  // clamped as doubles first: the conversion of a value outside int's range is undefined (NaN lands on 0)
  x0 = (int)fmin(fmax(kp[0] * sx - ps / 2.0, -1.0), (double)w);
  y0 = (int)fmin(fmax(kp[1] * sy - ps / 2.0, -1.0), (double)h);
  x0 = min(max(x0, 0), w - ps - 1);
  y0 = min(max(y0, 0), h - ps - 1);
}

// ps <= 16 (patch side: 16 by default, 10 / 8 in the reference's lighter configurations), C multiple of 16.
// 256 threads: 16 x-lanes x 16 channel groups for the loads, 16 pixels x 16 channel groups for the stores.
template <typename SRC, typename DST, int C>
__global__ __launch_bounds__(256) void extract_kernel(const SRC* __restrict__ fmap, int h, int w,
                                                      const double* __restrict__ kps, double sx, double sy,
                                                      int l2_normalize, DST* __restrict__ out,
                                                      int32_t* __restrict__ corners, double* __restrict__ scales,
                                                      int64_t first, int ps, const int* __restrict__ order) {
  constexpr int PSM = 16, CP = C + 1;           // +1 float of padding: conflict-free column reads
  __shared__ float tile[2 * PSM * CP];
  const int64_t k = order ? order[blockIdx.x] : blockIdx.x;   // patches in map-tile order, outputs at their own slots
  const int tid = threadIdx.x;
  int x0, y0;
  ex_corner(kps + 2 * k, sx, sy, ps, w, h, x0, y0);
  if (tid == 0) {
    corners[2 * (first + k)] = x0; corners[2 * (first + k) + 1] = y0;
    scales[2 * (first + k)] = sx; scales[2 * (first + k) + 1] = sy;
  }
  DST* patch = out + (size_t)(first + k) * ps * ps * C;
  const size_t plane = (size_t)h * w;
  const int lx = min(tid & 15, ps - 1), lc = tid >> 4;   // load mapping: 16 x-contiguous lanes (clamped to the patch), 16 channels per pass
  const int px = tid >> 4, sub = tid & 15;      // store mapping: 16 lanes per pixel, C/16 channels per lane
  constexpr int CPL = C / 16;
  // software pipeline: the CHW loads of row y + 1 are in flight while row y goes through the
  // (double-buffered) LDS tile, the normalisation and the HWC stores; one barrier per row
  SRC cur[CPL], nxt[CPL];
  {
    const SRC* row = fmap + (size_t)y0 * w + x0 + lx;
#pragma unroll
    for (int j = 0; j < CPL; ++j) cur[j] = row[(size_t)(lc + 16 * j) * plane];
  }
  for (int y = 0; y < ps; ++y) {
    float* tl = tile + (y & 1) * (PSM * CP);
#pragma unroll
    for (int j = 0; j < CPL; ++j) tl[lx * CP + lc + 16 * j] = (float)cur[j];
    if (y + 1 < ps) {
      const SRC* row = fmap + (size_t)(y0 + y + 1) * w + x0 + lx;
#pragma unroll
      for (int j = 0; j < CPL; ++j) nxt[j] = row[(size_t)(lc + 16 * j) * plane];
    }
    __syncthreads();
    float v[CPL];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { v[j] = tl[px * CP + sub * CPL + j]; ss = fmaf(v[j], v[j], ss); }
    if (l2_normalize) {   // torch.nn.functional.normalize(dim=1): v / max(||v||_2, 1e-12), fp32
      for (int off = 8; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
      const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int j = 0; j < CPL; ++j) v[j] = v[j] / den;
    }
    typedef DST vec_t __attribute__((ext_vector_type(CPL)));   // one lane's share of a texel: a single 16-B store at fp16
    vec_t o;
#pragma unroll
    for (int j = 0; j < CPL; ++j) o[j] = (DST)v[j];
    if (px < ps) *reinterpret_cast<vec_t*>(patch + ((size_t)y * ps + px) * C + sub * CPL) = o;
#pragma unroll
    for (int j = 0; j < CPL; ++j) cur[j] = nxt[j];
  }
}

// Few channels (the reference's weight-free `image` features, features/models/image.py: RGB or grey values; 3 / 1 channels):
// one thread per patch texel, its C channels from the C planes of the map; same corner / normalisation / cast as above.
template <typename SRC, typename DST, int C>
__global__ __launch_bounds__(256) void extract_small_kernel(const SRC* __restrict__ fmap, int h, int w,
                                                            const double* __restrict__ kps, double sx, double sy,
                                                            int l2_normalize, DST* __restrict__ out,
                                                            int32_t* __restrict__ corners, double* __restrict__ scales,
                                                            int64_t first, int ps) {
  const int64_t k = blockIdx.x;
  const int tid = threadIdx.x;
  int x0, y0;
  ex_corner(kps + 2 * k, sx, sy, ps, w, h, x0, y0);
  if (tid == 0) {
    corners[2 * (first + k)] = x0; corners[2 * (first + k) + 1] = y0;
    scales[2 * (first + k)] = sx; scales[2 * (first + k) + 1] = sy;
  }
  const size_t plane = (size_t)h * w;
  for (int t = tid; t < ps * ps; t += blockDim.x) {
    const int y = t / ps, x = t - y * ps;
    float v[C], ss = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { v[c] = (float)fmap[c * plane + (size_t)(y0 + y) * w + x0 + x]; ss = fmaf(v[c], v[c], ss); }
    if (l2_normalize) {
      const float den = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int c = 0; c < C; ++c) v[c] = v[c] / den;
    }
    DST* o = out + ((size_t)(first + k) * ps * ps + t) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) o[c] = (DST)v[c];
  }
}

// ---- map-tile order of the patches: histogram, scan, scatter ---------------------------------------------------------
__global__ __launch_bounds__(256) void ex_tile_count(int64_t n, const double* __restrict__ kps, double sx, double sy, int ps,
                                                     int w, int h, int tiles_x, int* __restrict__ tile_of, int* __restrict__ hist) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int x0, y0;
  ex_corner(kps + 2 * k, sx, sy, ps, w, h, x0, y0);
  const int t = (y0 >> 4) * tiles_x + (x0 >> 4);
  tile_of[k] = t;
  atomicAdd(hist + t, 1);
}

__global__ __launch_bounds__(1024) void ex_tile_scan(int n_tiles, int* __restrict__ hist /* counts -> first slots */) {
  __shared__ int part[1024];
  const int per = (n_tiles + 1023) / 1024, t0 = threadIdx.x * per;
  int s = 0;
  for (int i = t0; i < min(n_tiles, t0 + per); ++i) s += hist[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; } }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = t0; i < min(n_tiles, t0 + per); ++i) { const int v = hist[i]; hist[i] = run; run += v; }
}

__global__ __launch_bounds__(256) void ex_tile_scatter(int64_t n, const int* __restrict__ tile_of, int* __restrict__ cursor,
                                                       int* __restrict__ order) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  order[atomicAdd(cursor + tile_of[k], 1)] = (int)k;
}

}  // namespace pxr

extern "C" int pxr_arena_extract(pxr_ctx* ctx, pxr_arena* a, int64_t first, int64_t n, const void* d_fmap,
                                 int src_dtype, int h, int w, const double* d_keypoints, double image_w,
                                 double image_h, int l2_normalize) {
  using namespace pxr;
  PXR_REQUIRE(ctx && a && d_fmap && d_keypoints, "pxr_arena_extract: NULL argument");
  PXR_REQUIRE(first >= 0 && n >= 0 && first + n <= a->n, "pxr_arena_extract: range [%lld, %lld) outside arena of %lld patches",
              (long long)first, (long long)(first + n), (long long)a->n);
  PXR_REQUIRE(a->H == a->W && a->H >= 1 && a->H <= 16, "pxr_arena_extract: patch size %dx%d not supported (square, <= 16)", a->H, a->W);
  PXR_REQUIRE(h > a->H && w > a->W, "pxr_arena_extract: feature map %dx%d must exceed the patch size", h, w);
  PXR_REQUIRE(image_w > 0 && image_h > 0, "pxr_arena_extract: image size must be positive");
  if (n == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
  const double sx = (double)w / image_w, sy = (double)h / image_h;   // extractor.py:177
  // map-tile order of the patches (worth it once the patches outnumber the tiles a few times over)
  const int* d_order = nullptr;
  const int tiles_x = (w + 15) / 16, n_tiles = tiles_x * ((h + 15) / 16);
  if (n >= 4096 && n < ((int64_t)1 << 31)) {
    const size_t bytes = sizeof(int) * (2 * (size_t)n + n_tiles) + 768;
    if (bytes > ctx->workspace_bytes) {
      PXR_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->d_workspace) { PXR_HIP(hipFree(ctx->d_workspace)); ctx->d_workspace = nullptr; ctx->workspace_bytes = 0; }
      PXR_HIP(hipMalloc(&ctx->d_workspace, bytes));
      ctx->workspace_bytes = bytes;
    }
    int* tile_of = static_cast<int*>(ctx->d_workspace);
    int* order = tile_of + n;
    int* hist = order + n;
    PXR_HIP(hipMemsetAsync(hist, 0, sizeof(int) * n_tiles, ctx->stream));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(ex_tile_count, dim3(nb), dim3(256), 0, ctx->stream, n, d_keypoints, sx, sy, a->H, w, h, tiles_x, tile_of, hist);
    hipLaunchKernelGGL(ex_tile_scan, dim3(1), dim3(1024), 0, ctx->stream, n_tiles, hist);
    hipLaunchKernelGGL(ex_tile_scatter, dim3(nb), dim3(256), 0, ctx->stream, n, tile_of, hist, order);
    d_order = order;
  }
#define EX_LAUNCH(SRC, DST, CC)                                                                              \
  hipLaunchKernelGGL((extract_kernel<SRC, DST, CC>), dim3((unsigned)n), dim3(256), 0, ctx->stream,          \
                     (const SRC*)d_fmap, h, w, d_keypoints, sx, sy, l2_normalize, (DST*)a->d_data,          \
                     a->d_corners, a->d_scales, first, a->H, d_order)
#define EX_DST(SRC, CC)                                                   \
  do {                                                                    \
    if (a->dtype == PXR_F16) EX_LAUNCH(SRC, _Float16, CC);                \
    else if (a->dtype == PXR_F32) EX_LAUNCH(SRC, float, CC);              \
    else EX_LAUNCH(SRC, double, CC);                                      \
  } while (0)
  if (src_dtype == PXR_F32 && a->C == 128) EX_DST(float, 128);
  else if (src_dtype == PXR_F32 && a->C == 64) EX_DST(float, 64);
  else if (src_dtype == PXR_F16 && a->C == 128) EX_DST(_Float16, 128);
  else if (src_dtype == PXR_F16 && a->C == 64) EX_DST(_Float16, 64);
  else if ((src_dtype == PXR_F32 || src_dtype == PXR_F16) && (a->C == 3 || a->C == 1)) {
#define EX_SMALL(SRC, DST, CC)                                                                               \
  hipLaunchKernelGGL((extract_small_kernel<SRC, DST, CC>), dim3((unsigned)n), dim3(256), 0, ctx->stream,    \
                     (const SRC*)d_fmap, h, w, d_keypoints, sx, sy, l2_normalize, (DST*)a->d_data,          \
                     a->d_corners, a->d_scales, first, a->H)
#define EX_SMALL_DST(SRC, CC)                                             \
  do {                                                                    \
    if (a->dtype == PXR_F16) EX_SMALL(SRC, _Float16, CC);                 \
    else if (a->dtype == PXR_F32) EX_SMALL(SRC, float, CC);               \
    else EX_SMALL(SRC, double, CC);                                       \
  } while (0)
    if (src_dtype == PXR_F32 && a->C == 3) EX_SMALL_DST(float, 3);
    else if (src_dtype == PXR_F32) EX_SMALL_DST(float, 1);
    else if (a->C == 3) EX_SMALL_DST(_Float16, 3);
    else EX_SMALL_DST(_Float16, 1);
#undef EX_SMALL_DST
#undef EX_SMALL
  }
  else return set_error(PXR_EUNSUPPORTED, "pxr_arena_extract: source dtype %d / CHANNELS %d not supported (f16/f32 x 128/64/3/1)", src_dtype, a->C);
#undef EX_DST
#undef EX_LAUNCH
  return hip_check(hipGetLastError(), "extract_kernel launch");
}

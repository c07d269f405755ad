// pxr_ba_gram.hip -- the LM loop's evaluation of the feature-reference residual blocks from a CACHE of per-observation
// Gram matrices (gfx950).
//
// The exact-order kernel (pxr_ba_eval.hip, the statement of FeatureReferenceCostFunctor + BiCubicInterpolator,
// bundle_adjustment/src/../residuals/feature_reference.h:70-140, base/src/interpolation.h:177-218) reads the 4 x 4 x C stencil
// of every observation at every evaluation: 4 KB per observation, 4.9 GB per pass at configs[2], the HBM roofline of the LM
// iteration.  But the solver consumes only the 64-byte RECORD of a block -- |r|^2, the 2 x 2 Gram matrix of its image-space
// Jacobian and J^t r -- and bicubic interpolation is linear in the sixteen texels (pxr_gram.h): with G = T T^t and D = T d the
// record is six quadratic and three linear forms in the Catmull-Rom weights of the fractional position.  G and D depend on
// (patch, cell, reference) only, and an LM step moves a projection by a fraction of a texel: the cell rarely changes.  So
//   k_gram_eval<false>  projects every observation (WorldToPixel, base/src/projection.h:60-75 + FeaturePatch::ToPixelCoordinates,
//                       featurepatch.h:250-255); where the projection is still in the cell of the cached matrices it turns
//                       (G, D, weights) into the record -- 1.4 KB instead of 4 KB per observation and pass, no texel touched --
//                       and where it is not (every observation at the first evaluation of a solve) it lists the observation;
//   k_gram_build        builds G (32 v_mfma_f64_16x16x4: fp16 products are exact, the sums carry full double precision) and D
//                       for the listed observations: 1 408 bytes per observation in HBM (ten 4 x 4 blocks of the upper
//                       triangle + D);
//   k_gram_eval<true>   the records of the listed observations.
// Arithmetic: exact-in-fp64 algebra on G, where the reference interpolates with an fp32 horizontal pass
// (cubic_hermite_spline_simd.h) -- a record differs from the exact-order kernel's by that pass's own rounding (~1e-7 of the
// descriptor norm per channel, DESIGN.md section 13 has the measured differences); pxr_ba_eval -- the metric of bench.py, the
// parity tests of the functors -- stays the exact-order kernel.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "pxr_ba_solve.h"
#include "pxr_device.h"
#include "pxr_gram.h"
#include "pxr_internal.h"

namespace pxr {

constexpr int GC_STRIDE = IG_GDOUBLES + 16;      // doubles per observation in the cache: ten 4 x 4 blocks of G, then D
// (1 408 bytes = 128 mod 256: the eight observations of a wavefront alias in the LDS banks, 54 % of k_gram_eval's LDS cycles are
// conflicts.  Tried in round 5: an LDS image padded to 89 sixteen-byte pieces per observation -- the lanes of global_load_lds may ask
// for any 16 bytes -- removes them but needs a twelfth request per wavefront and an index computation per lane: 0.32 ms per pass
// instead of 0.27 ms.  The kernel waits on HBM, not on LDS; left as it is.)

struct GramArgs {
  pxr_ba_view v;
  const void* arena; const int32_t* corners; const double* scales;
  int H, W, l2_normalize;
  double* G;            // [n_obs][GC_STRIDE]
  int2* cell;           // [n_obs] (row, col) the cached matrices were built for
  int* dirty;           // [n_obs] 1: the projection left that cell at this evaluation -- rebuild, then evaluate (flags, not a
                        // compacted list: one atomic per wavefront on a shared counter serialised for > 1 ms when most moved)
  const double* r2;     // [n_points] d.d of the reference descriptors
  double* rec;
};

// One wavefront per workgroup, 64 consecutive observations at a time; the dirty ones go through a ring of three stencils in
// flight (a build's MFMA chain is shorter than the latency of its texels).
template <typename ST, int C>
__global__ __launch_bounds__(64) void k_gram_build(const GramArgs a) {
  __shared__ double refbuf[3][C];
  const int lane = threadIdx.x;
  const ST* arena = reinterpret_cast<const ST*>(a.arena);
  const size_t patch_elems = (size_t)a.H * a.W * C;
  const int64_t n = a.v.n_obs;
  for (int64_t base = (int64_t)blockIdx.x * 64; base < n; base += (int64_t)gridDim.x * 64) {
    const int64_t o = min(base + lane, n - 1);
    unsigned long long todo = __ballot(base + lane < n && a.dirty[o] != 0);
    if (todo == 0ull) continue;
    const int64_t pidx = a.v.d_obs_patch[o];
    const int pt = a.v.d_obs_point[o];
    const int2 cl = a.cell[o];
    const int pi_lo = (int)(pidx & 0xffffffffll), pi_hi = (int)(pidx >> 32);
    double2 rr0 = make_double2(0.0, 0.0), rr1 = rr0, rr2 = rr0;
    auto take = [&](GramTexels<ST, C>& tx, double2& rr) -> int {    // request the next dirty observation's texels and reference
      if (todo == 0ull) return -1;
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const int row = __builtin_amdgcn_readlane(cl.x, j), col = __builtin_amdgcn_readlane(cl.y, j);
      const int64_t pi = ((int64_t)__builtin_amdgcn_readlane(pi_hi, j) << 32) | (unsigned)__builtin_amdgcn_readlane(pi_lo, j);
      const int p = __builtin_amdgcn_readlane(pt, j);
#ifdef PXR_GRAM_PROBE_NO_LOAD      // tools/variant_build.sh: every build reads the same stencil (what the texel traffic costs)
      tx.load(arena, a.H, a.W, 5, 5);
#else
      tx.load(arena + (size_t)pi * patch_elems, a.H, a.W, row, col);
#endif
      if (2 * lane < C) rr = *reinterpret_cast<const double2*>(a.v.d_refs + (size_t)p * C + 2 * lane);
      return j;
    };
    auto build = [&](const GramTexels<ST, C>& tx, const double2& rr, int slot, int j) {
      if (2 * lane < C) *reinterpret_cast<double2*>(&refbuf[slot][2 * lane]) = rr;
      __syncthreads();
      double* g = a.G + (size_t)(base + j) * GC_STRIDE;
      gram_contract<ST, C>(tx, refbuf[slot], g, g + IG_GDOUBLES);
    };
    GramTexels<ST, C> t0, t1, t2;
    int s0 = take(t0, rr0), s1 = take(t1, rr1), s2 = take(t2, rr2);
    while (true) {
      if (s0 < 0) break;
      build(t0, rr0, 0, s0); s0 = take(t0, rr0);
      if (s1 < 0) break;
      build(t1, rr1, 1, s1); s1 = take(t1, rr1);
      if (s2 < 0) break;
      build(t2, rr2, 2, s2); s2 = take(t2, rr2);
    }
  }
}

// The records of up to eight observations by one wavefront, 8 lanes per observation (two rows of its G each); i: the observation
// of this lane's group (q < nq valid).  The 1 408 cached bytes of each observation are requested first, straight into LDS
// (global_load_lds_dwordx4: lane l of request j lands at S + 16 (64 j + l), no register in between; with CONTIGUOUS the eight
// observations are consecutive, 11 KB in one piece); the projection -- WorldToPixel, base/src/projection.h:60-75 +
// FeaturePatch::ToPixelCoordinates, featurepatch.h:250-255, on all eight lanes of the observation alike -- runs while they are
// on their way.  FIRST pass (every observation): one whose projection left the cell of its cached matrices is not evaluated
// but flagged (and the cell updated) for k_gram_build and the second pass.
template <bool FIRST>
__device__ __forceinline__ void gram_eval_eight(const GramArgs& a, double* S, int64_t i, int nq, int lane) {
  const int q = lane >> 3, sub = lane & 7, ri = sub >> 1, ci = 2 * (sub & 1);
  constexpr int NT = (8 * GC_STRIDE / 2 + 63) / 64, PER = GC_STRIDE / 2;
  static_assert(NT * 64 == 8 * PER, "the last request stays inside the buffer");
  const int n2 = nq * PER;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int e = min(lane + 64 * j, n2 - 1);
    const double2* src;
    if (FIRST) {
      src = reinterpret_cast<const double2*>(a.G + (size_t)__shfl((int)i, 0) * GC_STRIDE) + e;   // consecutive observations
    } else {
      const int64_t oi = __shfl((int)i, 8 * (e / PER));
      src = reinterpret_cast<const double2*>(a.G + (size_t)oi * GC_STRIDE) + e % PER;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(S + 2 * 64 * j), 16, 0, 0);
  }
  // -- the projection
  const int img = a.v.d_obs_image[i];
  const int pt = a.v.d_obs_point[i];
  const int64_t pidx = a.v.d_obs_patch[i];
  int2 cached = make_int2(0, 0);
  if (FIRST) cached = a.cell[i];
  const int cam = a.v.d_image_camera[img];
  double qv[4], t[3], X[3], k[PXR_KPAD];
#pragma unroll
  for (int j = 0; j < 4; ++j) qv[j] = a.v.d_qvec[4 * (size_t)img + j];
#pragma unroll
  for (int j = 0; j < 3; ++j) { t[j] = a.v.d_tvec[3 * (size_t)img + j]; X[j] = a.v.d_xyz[3 * (size_t)pt + j]; }
#pragma unroll
  for (int j = 0; j < PXR_KPAD; ++j) k[j] = a.v.d_cam_params[(size_t)cam * PXR_KPAD + j];
  const double sx = a.scales[2 * pidx], sy = a.scales[2 * pidx + 1];
  const double cx = (double)a.corners[2 * pidx], cy = (double)a.corners[2 * pidx + 1];
  const double r2 = a.r2[pt];
  double x, y;
  world_to_pixel(a.v.d_cam_model[cam], k, qv, t, X, x, y);
  const double u = x * sx - 0.5 - cx, v = y * sy - 0.5 - cy;
  const double rf = floor(v), cf = floor(u);
  const int row = texel_index(rf, a.H), col = texel_index(cf, a.W);
  const bool miss = FIRST && (cached.x != row || cached.y != col);
  if (FIRST && q < nq && sub == 0) {
    a.dirty[i] = miss ? 1 : 0;
    if (miss) a.cell[i] = make_int2(row, col);
  }
  double wu[4], dwu[4], wv[4], dwv[4];
  catmull_rom_weights(u - cf, wu, dwu);
  catmull_rom_weights(v - rf, wv, dwv);
  __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0): the matrices are in LDS (the compiler does not track these loads)
  __syncthreads();
  if (q >= nq || miss) return;
  const double* Gq = S + (size_t)q * GC_STRIDE;
  double ya[3], yb[3];                              // (G w, G wc, G wr) at rows 2 sub and 2 sub + 1
  gram_rows_times_weights(Gq, sub, wu, dwu, wv, dwv, ya, yb);
  const double2 d2 = *reinterpret_cast<const double2*>(Gq + IG_GDOUBLES + 2 * sub);
  const double wvo = pick4(wv, ri), dwvo = pick4(dwv, ri);
  const double wua = ci == 0 ? wu[0] : wu[2], wub = ci == 0 ? wu[1] : wu[3];
  const double dwua = ci == 0 ? dwu[0] : dwu[2], dwub = ci == 0 ? dwu[1] : dwu[3];
  const double oma = wvo * wua, omb = wvo * wub, omca = wvo * dwua, omcb = wvo * dwub, omra = dwvo * wua, omrb = dwvo * wub;
  const double Sgg = row8_sum(fma(oma, ya[0], omb * yb[0])), Sgc = row8_sum(fma(oma, ya[1], omb * yb[1]));
  const double Sgr = row8_sum(fma(oma, ya[2], omb * yb[2]));
  const double Scc = row8_sum(fma(omca, ya[1], omcb * yb[1])), Scr = row8_sum(fma(omca, ya[2], omcb * yb[2]));
  const double Srr = row8_sum(fma(omra, ya[2], omrb * yb[2]));
  const double Sfd = row8_sum(fma(oma, d2.x, omb * d2.y)), Scd = row8_sum(fma(omca, d2.x, omcb * d2.y));
  const double Srd = row8_sum(fma(omra, d2.x, omrb * d2.y));
  if (sub != 0) return;
  double s, gcc, gcr, grr, bc, br;
  if (a.l2_normalize) {       // r = f / |f| - d (interpolation.h:240-268: normalisation after the interpolation, chain rule)
    const double ninv = 1.0 / sqrt(Sgg), n2inv = ninv * ninv;
    const double pc = Sgc * n2inv, pr = Sgr * n2inv;
    s = 1.0 - 2.0 * Sfd * ninv + r2;
    gcc = (Scc - Sgc * pc) * n2inv; gcr = (Scr - Sgc * pr) * n2inv; grr = (Srr - Sgr * pr) * n2inv;
    bc = -(Scd - Sfd * pc) * ninv; br = -(Srd - Sfd * pr) * ninv;
  } else {                    // r = f - d
    s = Sgg - 2.0 * Sfd + r2;
    gcc = Scc; gcr = Scr; grr = Srr; bc = Sgc - Scd; br = Sgr - Srd;
  }
  if (s < 0.0) s = 0.0;       // (a NaN stays a NaN: a projection that cannot be evaluated rejects the step like in pxr_ba_eval)
  // Jet bridge: d/dx = dfdc * sx, d/dy = dfdr * sy  (interpolation.h:130-140 + featurepatch.h:250-255)
  double2* out = reinterpret_cast<double2*>(a.rec + (size_t)i * PXR_OBS_REC);
  out[0] = make_double2(s, gcc * sx * sx);
  out[1] = make_double2(gcr * sx * sy, grr * sy * sy);
  out[2] = make_double2(bc * sx, br * sy);
  out[3] = make_double2(x, y);
}

// first pass: observations 8 b .. 8 b + 7
__global__ __launch_bounds__(64) void k_gram_eval(const GramArgs a) {
  __shared__ __align__(16) double S[8 * GC_STRIDE];
  const int lane = threadIdx.x;
  const int64_t o0 = (int64_t)blockIdx.x * 8;
  const int nq = (int)min<int64_t>(8, a.v.n_obs - o0);
  if (nq <= 0) return;
  gram_eval_eight<true>(a, S, o0 + min(lane >> 3, nq - 1), nq, lane);
}

// second pass: the flagged ones of observations 64 b .. 64 b + 63, eight at a time
__global__ __launch_bounds__(64) void k_gram_eval_dirty(const GramArgs a) {
  __shared__ __align__(16) double S[8 * GC_STRIDE];
  const int lane = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * 64, n = a.v.n_obs;
  unsigned long long todo = __ballot(base + lane < n && a.dirty[min(base + lane, n - 1)] != 0);
#pragma clang loop unroll(disable)
  while (todo != 0ull) {
    const int nq = min(8, __popcll(todo));
    unsigned long long m = todo;                          // this lane group's observation: the (lane / 8)-th flagged one
    for (int t = min(lane >> 3, nq - 1); t > 0; --t) m &= m - 1ull;
    const int64_t i = base + __ffsll((long long)m) - 1;
    gram_eval_eight<false>(a, S, i, nq, lane);
    for (int t = 0; t < nq; ++t) todo &= todo - 1ull;
    __syncthreads();                                      // S is reused
  }
}

__global__ __launch_bounds__(256) void k_gram_count_dirty(int64_t n, const int* __restrict__ dirty, int* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long m = __ballot(i < n && dirty[min(i, n - 1)] != 0);
  if ((threadIdx.x & 63) == 0 && m != 0ull) atomicAdd(count, __popcll(m));
}

// d.d of every reference descriptor: 16 lanes per point
__global__ __launch_bounds__(256) void k_gram_r2(int64_t n_pts, int C, const double* __restrict__ refs, double* __restrict__ r2) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t p = t >> 4;
  const int sub = (int)(t & 15);
  double acc = 0.0;
  if (p < n_pts)
    for (int ch = sub; ch < C; ch += 16) { const double d = refs[(size_t)p * C + ch]; acc = fma(d, d, acc); }
  acc = row16_sum(acc);
  if (p < n_pts && sub == 0) r2[p] = acc;
}

__global__ __launch_bounds__(256) void k_gram_invalidate(int64_t n, int2* __restrict__ cell) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cell[i] = make_int2(-1000000, -1000000);
}

bool gram_eval_supported(const pxr_arena* arena, const pxr_ba_view* view) {
  return view->d_refs != nullptr && (arena->C == 128 || arena->C == 64) && (arena->dtype == PXR_F16 || arena->dtype == PXR_F32) &&
         arena->up == 1.0 && view->n_obs > 0 && view->n_obs < (int64_t)1 << 31;
}

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

static void gram_layout(void* base, int64_t n, int64_t np, GramCache* out, size_t* total) {
  const size_t bG = align256(sizeof(double) * (size_t)n * GC_STRIDE), bcell = align256(sizeof(int2) * (size_t)n),
               blist = align256(sizeof(int) * (size_t)n), bcount = 256, br2 = align256(sizeof(double) * (size_t)np);
  char* p = static_cast<char*>(base);
  out->G = reinterpret_cast<double*>(p); p += bG;
  out->cell = p; p += bcell;
  out->list = reinterpret_cast<int*>(p); p += blist;
  out->count = reinterpret_cast<int*>(p); p += bcount;
  out->r2 = reinterpret_cast<double*>(p);
  out->n_obs = n;
  *total = bG + bcell + blist + bcount + br2;
}

int gram_eval_prepare(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, GramCache* out) {
  const int64_t n = view->n_obs, np = view->n_points;
  size_t need = 0;
  gram_layout(nullptr, n, np, out, &need);
  if (ctx->gram_bytes < need) {          // grow-only, like the KA workspace: no hipMalloc per solve
    if (ctx->d_gram) PXR_HIP(hipFree(ctx->d_gram));
    ctx->d_gram = nullptr; ctx->gram_bytes = 0;
    // (PXR_GRAM_FAIL_ALLOC=1: the tests' way of running out of memory here)
    if (getenv("PXR_GRAM_FAIL_ALLOC") != nullptr || hipMalloc(&ctx->d_gram, need) != hipSuccess) { (void)hipGetLastError(); return set_error(PXR_ENOMEM, "Gram-matrix cache: %zu bytes", need); }
    ctx->gram_bytes = need;
  }
  gram_layout(ctx->d_gram, n, np, out, &need);
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(k_gram_invalidate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, static_cast<int2*>(out->cell));
  hipLaunchKernelGGL(k_gram_r2, dim3((unsigned)((np * 16 + 255) / 256)), dim3(256), 0, st, np, arena->C, view->d_refs, out->r2);
  return hip_check(hipGetLastError(), "Gram-matrix cache set-up");
}

// rebuild the matrices of the observations flagged in the cache's `dirty` array (the caller set the flags and the cells)
int gram_build_flagged(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* v, const GramCache& gc) {
  GramArgs a;
  a.v = *v;
  a.arena = arena->d_data; a.corners = arena->d_corners; a.scales = arena->d_scales;
  a.H = arena->H; a.W = arena->W; a.l2_normalize = 0;
  a.G = gc.G; a.cell = static_cast<int2*>(gc.cell); a.dirty = gc.list; a.r2 = gc.r2; a.rec = nullptr;
  const int64_t n = v->n_obs;
  const unsigned build_grid = (unsigned)std::min<int64_t>((n + 63) / 64, (int64_t)ctx->num_cus * 16);
#define GRAM_BUILD(ST, CC) hipLaunchKernelGGL((k_gram_build<ST, CC>), dim3(build_grid), dim3(64), 0, ctx->stream, a)
  if (arena->dtype == PXR_F16) { if (arena->C == 128) GRAM_BUILD(_Float16, 128); else GRAM_BUILD(_Float16, 64); }
  else { if (arena->C == 128) GRAM_BUILD(float, 128); else GRAM_BUILD(float, 64); }
#undef GRAM_BUILD
  return hip_check(hipGetLastError(), "Gram-matrix build");
}

int gram_evaluate(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* v, const pxr_interp_cfg* cfg, const GramCache& gc, double* rec) {
  GramArgs a;
  a.v = *v;
  a.arena = arena->d_data; a.corners = arena->d_corners; a.scales = arena->d_scales;
  a.H = arena->H; a.W = arena->W; a.l2_normalize = cfg->l2_normalize;
  a.G = gc.G; a.cell = static_cast<int2*>(gc.cell); a.dirty = gc.list; a.r2 = gc.r2; a.rec = rec;
  hipStream_t st = ctx->stream;
  const int64_t n = v->n_obs;
  hipLaunchKernelGGL(k_gram_eval, dim3((unsigned)((n + 7) / 8)), dim3(64), 0, st, a);
  // the observations that left their cell: rebuild, then evaluate them (both scan the flags: a wavefront with nothing to do leaves
  // after one 256-byte load)
  const unsigned build_grid = (unsigned)std::min<int64_t>((n + 63) / 64, (int64_t)ctx->num_cus * 16);
#define GRAM_BUILD(ST, CC) hipLaunchKernelGGL((k_gram_build<ST, CC>), dim3(build_grid), dim3(64), 0, st, a)
  if (arena->dtype == PXR_F16) { if (arena->C == 128) GRAM_BUILD(_Float16, 128); else GRAM_BUILD(_Float16, 64); }
  else { if (arena->C == 128) GRAM_BUILD(float, 128); else GRAM_BUILD(float, 64); }
#undef GRAM_BUILD
  hipLaunchKernelGGL(k_gram_eval_dirty, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, a);
  return hip_check(hipGetLastError(), "Gram-matrix evaluation");
}

}  // namespace pxr

// C-ABI: the records of pxr_ba_eval(with_jacobian = 1) through the Gram-matrix cache, outside a solve (parity tests, bench).
extern "C" int pxr_ba_eval_gram(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg, int reset,
                                double* d_rec, int32_t* h_rebuilt) {
  using namespace pxr;
  PXR_REQUIRE(ctx && arena && view && cfg && d_rec, "pxr_ba_eval_gram: NULL argument");
  PXR_REQUIRE(gram_eval_supported(arena, view), "pxr_ba_eval_gram: needs feature patches of 128 / 64 channels in fp16 / fp32 storage and reference descriptors");
  PXR_HIP(hipSetDevice(ctx->device));
  GramCache gc;
  size_t need = 0;
  gram_layout(ctx->d_gram, view->n_obs, view->n_points, &gc, &need);
  if (reset || ctx->gram_bytes < need) {
    if (int rc = gram_eval_prepare(ctx, arena, view, &gc)) return rc;
  }
  if (int rc = gram_evaluate(ctx, arena, view, cfg, gc, d_rec)) return rc;
  if (h_rebuilt) {
    PXR_HIP(hipMemsetAsync(gc.count, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_gram_count_dirty, dim3((unsigned)((view->n_obs + 255) / 256)), dim3(256), 0, ctx->stream, view->n_obs, gc.list, gc.count);
    PXR_HIP(hipMemcpyAsync(h_rebuilt, gc.count, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    PXR_HIP(hipStreamSynchronize(ctx->stream));
  }
  return PXR_OK;
}


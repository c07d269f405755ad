// pxr_internal.h -- host-side internals of libpixsfm_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>

#include "pixsfm_hip.h"

struct pxr_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  hipEvent_t ev_sync = nullptr;  // the LM loop's polled host synchronisation (pxr_ba_solve)
  void* h_readback = nullptr;    // 4 KiB of pinned host memory: the LM loop's per-attempt read-back (an async copy into pageable memory blocks the host)
  void* h_setup = nullptr;       // grow-only pinned staging of the solvers' set-up (per-point offsets for the host's table building)
  size_t h_setup_bytes = 0;
  double* d_scratch = nullptr;   // small reduction scratch (device)
  size_t scratch_bytes = 0;
  int num_cus = 256;
  void* d_workspace = nullptr;   // grow-only solver workspace (KA), reused across calls
  size_t workspace_bytes = 0;
  void* d_workspace_mat = nullptr;   // grow-only storage of the KA normal-matrix blocks
  size_t workspace_mat_bytes = 0;
  void* comm = nullptr;          // ncclComm_t of this rank (pxr_comm.cpp), NULL on a single GPU
  int rank = 0, nranks = 1;
  bool force_collective = false; // pxr_comm_force / PXR_FORCE_COLLECTIVE=1: a ONE-rank communicator still runs every collective of the solvers through RCCL
  int64_t collective_calls = 0;  // ncclAllReduce calls issued through this context (diagnostics: pxr_comm_stats)
  int64_t collective_bytes = 0;
  pxr_iteration_callback iter_cb = nullptr;   // pxr_set_iteration_callback
  void* iter_user = nullptr;
  void* h_stage[2] = {nullptr, nullptr};      // pinned staging buffers of the patch uploads (pxr_arena_upload*), lazily allocated
  hipEvent_t ev_stage[2] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  bool deterministic = true;     // pxr_set_deterministic / PXR_DETERMINISTIC=0 opts out: order- and partition-independent accumulation in the solvers
  bool gram_cache = true;        // pxr_set_gram_cache / PXR_GRAM_CACHE=0 opts out: pxr_ba_solve evaluates from cached Gram matrices (pxr_ba_gram.hip)
  void* d_gram = nullptr;        // grow-only storage of that cache
  size_t gram_bytes = 0;
  void* d_solve_arena = nullptr; // grow-only storage of pxr_ba_solve's ~60 work buffers (sized by the previous solve; PXR_BA_ARENA=0: hipMalloc / hipFree per buffer)
  size_t solve_arena_bytes = 0;
};

struct pxr_arena {
  pxr_ctx* ctx = nullptr;
  int dtype = PXR_F16, C = 0, H = 0, W = 0;
  int64_t n = 0;
  void* d_data = nullptr;
  bool owns_data = false;
  int32_t* d_corners = nullptr;
  double* d_scales = nullptr;
  double up = 1.0;               // FeaturePatch::upsampling_factor_ of every patch (cost maps; 1 for feature patches)
  size_t elem_size() const { return dtype == PXR_F16 ? 2 : (dtype == PXR_F32 ? 4 : 8); }
  size_t patch_bytes() const { return (size_t)H * W * C * elem_size(); }
};

namespace pxr {
int set_error(int code, const char* fmt, ...);
inline int hip_check(hipError_t e, const char* what) {
  if (e == hipSuccess) return PXR_OK;
  return set_error(PXR_EHIP, "%s: %s", what, hipGetErrorString(e));
}
// `bytes` of pinned host memory owned by the context (grow-only; NULL when it cannot be had: the caller takes its pageable path)
void* setup_staging(pxr_ctx* ctx, size_t bytes);
// in-place all-reduce(sum) on the context's stream through its RCCL communicator (no-op without one)
int comm_allreduce_sum(pxr_ctx* ctx, double* d_buf, int64_t count, bool even_single_rank = false);
int comm_allreduce_sum_i64(pxr_ctx* ctx, long long* d_buf, int64_t count);
// pxr_ba_eval with the cost reduction fused into the residual kernel: *d_cost_sum += sum 0.5 rho(|r|^2)
int ba_eval_with_cost(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, const pxr_interp_cfg* cfg,
                      int with_jacobian, double* d_rec, double* d_r, double* d_gx, double* d_gy,
                      const pxr_loss* loss, double* d_cost_sum);
// pxr_ba_inner.hip: the inner iterations' split of the points by track length (made once per solve)
struct InnerLists {
  void* d_short = nullptr; int64_t n_short = 0; int maxo_short = 1;  // points with 1 .. 16 observations (Gram-matrix kernel): {point, length, first slot}
  int* d_long = nullptr; int64_t n_long = 0;                          // the others (packed kernel, one point per wavefront)
  void* d_slots = nullptr;                                            // the Gram-matrix kernel's table: per listed point {point, length, first slot} + {image, camera, patch, observation} x maxo_short
  void* d_waves = nullptr; int64_t n_waves = 0;                       // the packed Gram-matrix kernel's table: per wavefront {<= 4 points, first slots} + 16 observation slots
  void* d_wave_heads = nullptr;
  bool own_short = true, own_long = true, own_slots = true, own_heads = true;   // false: carved out of the solve's arena (solve_scratch), not freed
};
// pxr_ba_solve.hip: `bytes` of device memory for the duration of the running pxr_ba_solve -- from the context's arena when it has
// room (*owned = false), else from hipMalloc (*owned = true: the caller frees it)
void* solve_scratch(size_t bytes, bool* owned);
// pxr_ba_gram.hip: the per-observation Gram matrices of one solve (storage owned by the context)
struct GramCache {
  double* G = nullptr;       // [n_obs][176]: ten 4 x 4 blocks of the upper triangle of G = T T^t, then D = T ref
  void* cell = nullptr;      // [n_obs] int2 (row, col): the 4 x 4 cell the observation's matrices were built for
  int* list = nullptr;       // [n_obs] flags: 1 = the projection left that cell, rebuild (k_gram_eval / k_gram_flag_slots set them)
  int* count = nullptr;      // one counter (pxr_ba_eval_gram's h_rebuilt)
  double* r2 = nullptr;      // [n_points] d.d of the reference descriptors
  int64_t n_obs = 0;
};
bool gram_eval_supported(const pxr_arena* arena, const pxr_ba_view* view);
int gram_eval_prepare(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* view, GramCache* out);
// the records of pxr_ba_eval(with_jacobian = 1) at the parameters of `v`, from the cache (rebuilding what moved to another cell)
int gram_evaluate(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* v, const pxr_interp_cfg* cfg, const GramCache& gc, double* rec);
// rebuild the matrices of the observations flagged in gc.list (1: rebuild; the caller also updated gc.cell)
int gram_build_flagged(pxr_ctx* ctx, pxr_arena* arena, const pxr_ba_view* v, const GramCache& gc);
}  // namespace pxr

#define PXR_HIP(call)                                        \
  do {                                                       \
    int _rc = pxr::hip_check((call), #call);                 \
    if (_rc != PXR_OK) return _rc;                           \
  } while (0)

#define PXR_REQUIRE(cond, ...)                               \
  do {                                                       \
    if (!(cond)) return pxr::set_error(PXR_EINVAL, __VA_ARGS__); \
  } while (0)

// pxr_comm.cpp -- the context's RCCL communicator: ncclAllReduce over xGMI on the
// context's HIP stream, so that the LM loop of pxr_ba_solve has no host-language hop in its
// collective.  librccl is resolved with dlopen at run time: first the copy the process already
// mapped (PyTorch ships its own librccl.so with the same soname), then the ROCm one.  Only the
// five entry points below are used; their prototypes are restated from rccl.h (ROCm 7.2,
// /opt/rocm/include/rccl/rccl.h:187-260,611) so the library builds without that header.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "pxr_internal.h"

namespace pxr {
namespace {

struct NcclUniqueId { char internal[PXR_COMM_ID_BYTES]; };   // ncclUniqueId, rccl.h:43
typedef void* NcclComm;                                      // ncclComm_t
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclInt64 = 4, kNcclFloat64 = 8 };   // ncclResult_t / ncclRedOp_t / ncclDataType_t values (rccl.h:459-467)

struct Rccl {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) {            // the copy this process already mapped, if any
      r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      if (r.handle) break;
    }
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (size_t i = 0; !r.handle && i < sizeof(paths) / sizeof(paths[0]); ++i) r.handle = dlopen(paths[i], RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) {
      const char* why = dlerror();             // ONE call: dlerror() clears the message it returns
      r.error = std::string("librccl not found: ") + (why ? why : "?");
      return;
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) r.error = "librccl lacks a required entry point";
  });
  return &r;
}

int nccl_check(int rc, const char* what) {
  if (rc == kNcclSuccess) return PXR_OK;
  Rccl* r = rccl();
  return set_error(PXR_EHIP, "%s: %s", what, r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
}

}  // namespace

int comm_allreduce_sum(pxr_ctx* ctx, double* d_buf, int64_t count, bool even_single_rank) {
  if (!ctx->comm || count <= 0 || (ctx->nranks <= 1 && !even_single_rank && !ctx->force_collective)) return PXR_OK;
  Rccl* r = rccl();
  ++ctx->collective_calls; ctx->collective_bytes += count * 8;
  return nccl_check(r->AllReduce(d_buf, d_buf, (size_t)count, kNcclFloat64, kNcclSum, (NcclComm)ctx->comm, ctx->stream),
                    "ncclAllReduce");
}

// the same for 64-bit integers: the deterministic solvers' fixed-point slots and scalar limbs (integer sums are associative, so
// an N-rank solve accumulates the very integers of the one-rank solve)
int comm_allreduce_sum_i64(pxr_ctx* ctx, long long* d_buf, int64_t count) {
  if (!ctx->comm || count <= 0 || (ctx->nranks <= 1 && !ctx->force_collective)) return PXR_OK;
  Rccl* r = rccl();
  ++ctx->collective_calls; ctx->collective_bytes += count * 8;
  return nccl_check(r->AllReduce(d_buf, d_buf, (size_t)count, kNcclInt64, kNcclSum, (NcclComm)ctx->comm, ctx->stream),
                    "ncclAllReduce(int64)");
}

}  // namespace pxr

extern "C" {

int pxr_comm_unique_id(void* h_id) {
  using namespace pxr;
  PXR_REQUIRE(h_id, "pxr_comm_unique_id: NULL buffer");
  Rccl* r = rccl();
  if (!r->error.empty()) return set_error(PXR_EUNSUPPORTED, "pxr_comm_unique_id: %s", r->error.c_str());
  NcclUniqueId id;
  int rc = nccl_check(r->GetUniqueId(&id), "ncclGetUniqueId");
  if (rc != PXR_OK) return rc;
  std::memcpy(h_id, id.internal, PXR_COMM_ID_BYTES);
  return PXR_OK;
}

int pxr_comm_init(pxr_ctx* ctx, const void* h_id, int rank, int nranks) {
  using namespace pxr;
  PXR_REQUIRE(ctx && h_id, "pxr_comm_init: NULL argument");
  PXR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "pxr_comm_init: rank %d of %d", rank, nranks);
  PXR_REQUIRE(ctx->comm == nullptr, "pxr_comm_init: the context already owns a communicator");
  Rccl* r = rccl();
  if (!r->error.empty()) return set_error(PXR_EUNSUPPORTED, "pxr_comm_init: %s", r->error.c_str());
  PXR_HIP(hipSetDevice(ctx->device));
  NcclUniqueId id;
  std::memcpy(id.internal, h_id, PXR_COMM_ID_BYTES);
  NcclComm comm = nullptr;
  int rc = nccl_check(r->CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank");
  if (rc != PXR_OK) return rc;
  ctx->comm = comm; ctx->rank = rank; ctx->nranks = nranks;
  return PXR_OK;
}

int pxr_comm_destroy(pxr_ctx* ctx) {
  using namespace pxr;
  PXR_REQUIRE(ctx, "pxr_comm_destroy: ctx is NULL");
  if (ctx->comm) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)rccl()->CommDestroy((NcclComm)ctx->comm);
    ctx->comm = nullptr;
  }
  ctx->rank = 0; ctx->nranks = 1;
  return PXR_OK;
}

int pxr_comm_set_rank(pxr_ctx* ctx, int rank, int nranks) {
  PXR_REQUIRE(ctx, "pxr_comm_set_rank: ctx is NULL");
  PXR_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "pxr_comm_set_rank: rank %d of %d", rank, nranks);
  PXR_REQUIRE(ctx->comm == nullptr, "pxr_comm_set_rank: the context owns a communicator (its rank is fixed)");
  ctx->rank = rank; ctx->nranks = nranks;
  return PXR_OK;
}

int pxr_comm_rank(pxr_ctx* ctx, int* rank, int* nranks) {
  PXR_REQUIRE(ctx, "pxr_comm_rank: ctx is NULL");
  if (rank) *rank = ctx->rank;
  if (nranks) *nranks = ctx->nranks;
  return PXR_OK;
}

int pxr_comm_force(pxr_ctx* ctx, int on) {
  PXR_REQUIRE(ctx, "pxr_comm_force: ctx is NULL");
  ctx->force_collective = on != 0;
  return PXR_OK;
}

int pxr_comm_stats(pxr_ctx* ctx, int64_t* calls, int64_t* bytes, int reset) {
  PXR_REQUIRE(ctx, "pxr_comm_stats: ctx is NULL");
  if (calls) *calls = ctx->collective_calls;
  if (bytes) *bytes = ctx->collective_bytes;
  if (reset) { ctx->collective_calls = 0; ctx->collective_bytes = 0; }
  return PXR_OK;
}

int pxr_comm_allreduce_sum(pxr_ctx* ctx, double* d_buf, int64_t count) {
  PXR_REQUIRE(ctx && (count == 0 || d_buf), "pxr_comm_allreduce_sum: NULL argument");
  PXR_REQUIRE(count >= 0, "pxr_comm_allreduce_sum: negative count");
  PXR_REQUIRE(ctx->nranks <= 1 || ctx->comm, "pxr_comm_allreduce_sum: %d ranks but no communicator (pxr_comm_init)", ctx->nranks);
  return pxr::comm_allreduce_sum(ctx, d_buf, count, true);   // a one-rank communicator still goes through RCCL here
}

}  // extern "C"

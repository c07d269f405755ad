// pxr_runtime.cpp -- context, error reporting, device-memory plumbing and the patch arena.
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "pxr_internal.h"

namespace pxr {
static thread_local std::string g_error;
int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}
void* setup_staging(pxr_ctx* ctx, size_t bytes) {
  if (ctx->h_setup_bytes >= bytes) return ctx->h_setup;
  if (ctx->h_setup) { (void)hipHostFree(ctx->h_setup); ctx->h_setup = nullptr; ctx->h_setup_bytes = 0; }
  const size_t want = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
  if (hipHostMalloc(&ctx->h_setup, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->h_setup = nullptr; return nullptr; }
  ctx->h_setup_bytes = want;
  return ctx->h_setup;
}
}  // namespace pxr

static void release_staging(pxr_ctx* ctx);

extern "C" {

int pxr_version(void) { return 100; }

const char* pxr_last_error(void) { return pxr::g_error.c_str(); }

int pxr_ctx_create(int device, void* stream, pxr_ctx** out) {
  PXR_REQUIRE(out != nullptr, "pxr_ctx_create: out is NULL");
  int count = 0;
  PXR_HIP(hipGetDeviceCount(&count));
  PXR_REQUIRE(device >= 0 && device < count, "pxr_ctx_create: device %d out of range (%d visible)",
              device, count);
  PXR_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PXR_HIP(hipGetDeviceProperties(&prop, device));
  pxr_ctx* c = new pxr_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->num_cus = prop.multiProcessorCount;
  PXR_HIP(hipEventCreate(&c->ev_start));
  PXR_HIP(hipEventCreate(&c->ev_stop));
  PXR_HIP(hipEventCreateWithFlags(&c->ev_sync, hipEventDisableTiming));
  if (const char* e = std::getenv("PXR_DETERMINISTIC")) c->deterministic = e[0] != '\0' && e[0] != '0';
  if (const char* e = std::getenv("PXR_GRAM_CACHE")) c->gram_cache = e[0] != '\0' && e[0] != '0';
  if (const char* e = std::getenv("PXR_FORCE_COLLECTIVE")) c->force_collective = e[0] != '\0' && e[0] != '0';
  c->scratch_bytes = 1 << 20;
  PXR_HIP(hipMalloc((void**)&c->d_scratch, c->scratch_bytes));
  *out = c;
  return PXR_OK;
}

int pxr_ctx_destroy(pxr_ctx* ctx) {
  if (!ctx) return PXR_OK;
  // teardown: nothing useful can be done with an error here
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->h_readback) { (void)hipHostFree(ctx->h_readback); ctx->h_readback = nullptr; }
  if (ctx->h_setup) { (void)hipHostFree(ctx->h_setup); ctx->h_setup = nullptr; ctx->h_setup_bytes = 0; }
  if (ctx->comm) (void)pxr_comm_destroy(ctx);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->d_workspace) (void)hipFree(ctx->d_workspace);
  if (ctx->d_workspace_mat) (void)hipFree(ctx->d_workspace_mat);
  if (ctx->d_gram) (void)hipFree(ctx->d_gram);
  if (ctx->d_solve_arena) (void)hipFree(ctx->d_solve_arena);
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
  if (ctx->ev_sync) (void)hipEventDestroy(ctx->ev_sync);
  release_staging(ctx);
  delete ctx;
  return PXR_OK;
}

int pxr_ctx_sync(pxr_ctx* ctx) {
  PXR_REQUIRE(ctx, "pxr_ctx_sync: ctx is NULL");
  PXR_HIP(hipStreamSynchronize(ctx->stream));
  return PXR_OK;
}

int pxr_malloc(pxr_ctx* ctx, size_t bytes, void** d_ptr) {
  PXR_REQUIRE(ctx && d_ptr, "pxr_malloc: NULL argument");
  PXR_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
  if (e == hipErrorOutOfMemory) return pxr::set_error(PXR_ENOMEM, "pxr_malloc: out of device memory (%zu bytes)", bytes);
  return pxr::hip_check(e, "hipMalloc");
}

int pxr_free(pxr_ctx* ctx, void* d_ptr) {
  PXR_REQUIRE(ctx, "pxr_free: ctx is NULL");
  if (d_ptr) PXR_HIP(hipFree(d_ptr));
  return PXR_OK;
}

int pxr_memcpy_h2d(pxr_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  PXR_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), "pxr_memcpy_h2d: NULL argument");
  if (bytes == 0) return PXR_OK;
  PXR_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  PXR_HIP(hipStreamSynchronize(ctx->stream));
  return PXR_OK;
}

int pxr_memcpy_d2h(pxr_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  PXR_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), "pxr_memcpy_d2h: NULL argument");
  if (bytes == 0) return PXR_OK;
  PXR_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PXR_HIP(hipStreamSynchronize(ctx->stream));
  return PXR_OK;
}

int pxr_memset(pxr_ctx* ctx, void* d_dst, int value, size_t bytes) {
  PXR_REQUIRE(ctx && (bytes == 0 || d_dst), "pxr_memset: NULL argument");
  if (bytes == 0) return PXR_OK;
  PXR_HIP(hipMemsetAsync(d_dst, value, bytes, ctx->stream));
  return PXR_OK;
}

int pxr_timer_start(pxr_ctx* ctx) {
  PXR_REQUIRE(ctx, "pxr_timer_start: ctx is NULL");
  PXR_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
  return PXR_OK;
}

int pxr_timer_stop(pxr_ctx* ctx, double* ms) {
  PXR_REQUIRE(ctx && ms, "pxr_timer_stop: NULL argument");
  PXR_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
  PXR_HIP(hipEventSynchronize(ctx->ev_stop));
  float f = 0.f;
  PXR_HIP(hipEventElapsedTime(&f, ctx->ev_start, ctx->ev_stop));
  *ms = (double)f;
  return PXR_OK;
}

// ---- arena ---------------------------------------------------------------------------
int pxr_arena_create(pxr_ctx* ctx, int dtype, int C, int H, int W, int64_t n_patches,
                     void* d_data_or_null, pxr_arena** out) {
  PXR_REQUIRE(ctx && out, "pxr_arena_create: NULL argument");
  PXR_REQUIRE(dtype == PXR_F16 || dtype == PXR_F32 || dtype == PXR_F64,
              "pxr_arena_create: unknown dtype %d", dtype);
  PXR_REQUIRE(C >= 1 && H >= 1 && W >= 1 && n_patches >= 0,
              "pxr_arena_create: invalid shape C=%d H=%d W=%d n=%lld", C, H, W, (long long)n_patches);
  PXR_HIP(hipSetDevice(ctx->device));
  pxr_arena* a = new pxr_arena();
  a->ctx = ctx; a->dtype = dtype; a->C = C; a->H = H; a->W = W; a->n = n_patches;
  const size_t bytes = a->patch_bytes() * (size_t)n_patches;
  if (d_data_or_null) {
    a->d_data = d_data_or_null;
    a->owns_data = false;
  } else {
    hipError_t e = hipMalloc(&a->d_data, bytes ? bytes : 1);
    if (e != hipSuccess) {
      delete a;
      return pxr::set_error(e == hipErrorOutOfMemory ? PXR_ENOMEM : PXR_EHIP,
                            "pxr_arena_create: hipMalloc(%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    a->owns_data = true;
  }
  const size_t n1 = n_patches ? (size_t)n_patches : 1;
  hipError_t e1 = hipMalloc((void**)&a->d_corners, n1 * 2 * sizeof(int32_t));
  hipError_t e2 = hipMalloc((void**)&a->d_scales, n1 * 2 * sizeof(double));
  if (e1 != hipSuccess || e2 != hipSuccess) {
    pxr_arena_destroy(a);
    return pxr::set_error(PXR_ENOMEM, "pxr_arena_create: metadata allocation failed");
  }
  *out = a;
  return PXR_OK;
}

int pxr_arena_destroy(pxr_arena* a) {
  if (!a) return PXR_OK;
  (void)hipSetDevice(a->ctx->device);
  (void)hipStreamSynchronize(a->ctx->stream);
  if (a->owns_data && a->d_data) (void)hipFree(a->d_data);
  if (a->d_corners) (void)hipFree(a->d_corners);
  if (a->d_scales) (void)hipFree(a->d_scales);
  delete a;
  return PXR_OK;
}

// Host patches -> arena through two pinned staging buffers: while the DMA engine drains one, the host cores gather the next
// chunk into the other (from ONE contiguous block, or from `count` separate patches -- the FeaturePatch objects of a
// FeatureManager: no stacked host copy of the whole set first).  A pageable hipMemcpyAsync is staged by the runtime one
// small buffer at a time on one thread; this keeps the link busy instead.
// Copy into the pinned staging buffer with streaming stores: a plain memcpy reads the destination lines before writing them
// (write-allocate), and while the copy engine drains the other staging buffer the host DRAM is the shared resource -- measured
// on the GPU box: 32 memcpy threads and the DMA slowed each other to 36 GB/s of a 57 GB/s link.
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void copy_streaming_avx2(char* dst, const char* src, size_t bytes) {
  size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
  if (head > bytes) head = bytes;
  std::memcpy(dst, src, head);
  dst += head; src += head; bytes -= head;
  const size_t body = bytes & ~(size_t)127;
  for (size_t o = 0; o < body; o += 128) {
    const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + o));
    const __m256i v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + o + 32));
    const __m256i v2 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + o + 64));
    const __m256i v3 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + o + 96));
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + o), v0);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + o + 32), v1);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + o + 64), v2);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + o + 96), v3);
  }
  std::memcpy(dst + body, src + body, bytes - body);
}
static bool streaming_copy_available() { return __builtin_cpu_supports("avx2"); }
static void streaming_fence() { _mm_sfence(); }
#else   // other hosts (aarch64, ppc64le ROCm boxes): plain memcpy
static void copy_streaming_avx2(char* dst, const char* src, size_t bytes) { std::memcpy(dst, src, bytes); }
static bool streaming_copy_available() { return false; }
static void streaming_fence() {}
#endif
static void copy_to_staging(char* dst, const char* src, size_t bytes, bool streaming) {
  if (streaming && bytes >= 4096) copy_streaming_avx2(dst, src, bytes);
  else std::memcpy(dst, src, bytes);
}

// hardware threads, capped by the cgroup CPU quota (v2: /sys/fs/cgroup/cpu.max "<quota> <period>" or "max ...")
static unsigned usable_cpus() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64]; long long period = 0;
    if (std::fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
      const long long quota = std::atoll(q);
      if (quota > 0) n = (unsigned)std::max<long long>(1, std::min<long long>(n, (quota + period - 1) / period));
    }
    std::fclose(f);
  }
  return n;
}

namespace {
// The gather threads of ONE upload call: created once, handed one byte range per staging chunk (a fresh std::thread per
// 256 MiB chunk cost ~250 creations for the 65.5 GB arena of BASELINE configs[2]).
class GatherPool {
 public:
  explicit GatherPool(unsigned n) {
    for (unsigned t = 0; t < n; ++t) threads_.emplace_back([this, t] { loop(t); });
  }
  ~GatherPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; ++epoch_; }
    cv_.notify_all();
    for (auto& th : threads_) th.join();
  }
  unsigned size() const { return (unsigned)threads_.size(); }
  // runs fn(t) on every thread t and returns when all are done
  void run(const std::function<void(unsigned)>& fn) {
    { std::lock_guard<std::mutex> g(m_); fn_ = &fn; pending_ = size(); ++epoch_; }
    cv_.notify_all();
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
  }
 private:
  void loop(unsigned t) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)>* fn;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
        fn = fn_;
      }
      (*fn)(t);
      { std::lock_guard<std::mutex> g(m_); if (--pending_ == 0) done_.notify_one(); }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned)>* fn_ = nullptr;
  unsigned pending_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};
}  // namespace

static void release_staging(pxr_ctx* ctx) {
  for (int q = 0; q < 2; ++q) {
    if (ctx->h_stage[q]) (void)hipHostFree(ctx->h_stage[q]);
    if (ctx->ev_stage[q]) (void)hipEventDestroy(ctx->ev_stage[q]);
    ctx->h_stage[q] = nullptr; ctx->ev_stage[q] = nullptr;
  }
  ctx->stage_bytes = 0;
}

// Two pinned buffers of min(256 MiB, the upload) bytes each (grown when a later upload is larger); any failure leaves the
// context without staging state.
static int ensure_staging(pxr_ctx* ctx, size_t total_bytes) {
  constexpr size_t kStage = (size_t)256 << 20, kMin = (size_t)1 << 20;
  size_t want = std::min(kStage, std::max(kMin, total_bytes));
  if (const char* e = std::getenv("PXR_UPLOAD_STAGE_BYTES")) want = std::max<size_t>(4096, (size_t)std::atoll(e));   // tests
  want = (want + 4095) & ~(size_t)4095;
  if (ctx->h_stage[0] && ctx->stage_bytes >= want) return PXR_OK;
  if (ctx->h_stage[0]) (void)hipStreamSynchronize(ctx->stream);        // an earlier upload may still read the old buffers
  release_staging(ctx);
  for (int b = 0; b < 2; ++b) {
    if (hipHostMalloc(&ctx->h_stage[b], want, hipHostMallocDefault) != hipSuccess) {
      ctx->h_stage[b] = nullptr;
      release_staging(ctx);
      return pxr::set_error(PXR_ENOMEM, "pxr_arena_upload: no pinned staging memory (2 x %zu bytes)", want);
    }
    if (hipEventCreateWithFlags(&ctx->ev_stage[b], hipEventDisableTiming) != hipSuccess) {
      ctx->ev_stage[b] = nullptr;
      release_staging(ctx);
      return pxr::set_error(PXR_EHIP, "pxr_arena_upload: hipEventCreate failed");
    }
  }
  ctx->stage_bytes = want;
  return PXR_OK;
}

// The upload is ONE byte stream of count * patch_bytes bytes cut into staging-buffer-sized chunks; a chunk may begin and end
// inside a patch, so a patch larger than the staging buffer (a dense 1600 x 1200 x 128 fp16 map is 491 MB) takes several.
static int upload_staged(pxr_arena* a, int64_t first, int64_t count, const void* h_contig, const void* const* h_ptrs) {
  pxr_ctx* ctx = a->ctx;
  hipStream_t s = ctx->stream;
  const size_t pb = a->patch_bytes();
  const size_t total = pb * (size_t)count;
  if (int rc = ensure_staging(ctx, total)) return rc;
  const size_t stage = ctx->stage_bytes;
  // 8 threads already keep the link busy (56 GB/s measured; 16 leave slack for slower hosts); PXR_UPLOAD_THREADS overrides.
  // Half of the CPUs the process can really use: a container's cgroup quota counts (the GPU boxes of this project: 256
  // hardware threads, quota 16 -- sixteen gather threads left nothing for the caller's own work beside a prefetch)
  unsigned n_thr = std::max(1u, std::min(usable_cpus() / 2, 16u));
  if (const char* e = std::getenv("PXR_UPLOAD_THREADS")) n_thr = (unsigned)std::max(1, std::atoi(e));
  if (total < ((size_t)4 << 20)) n_thr = 1;
  const bool streaming = streaming_copy_available() && std::getenv("PXR_UPLOAD_NO_STREAMING") == nullptr;
  auto source = [&](size_t patch) {
    return h_ptrs ? static_cast<const char*>(h_ptrs[patch]) : static_cast<const char*>(h_contig) + patch * pb;
  };
  // bytes [o0, o1) of the stream -> dst + (o0 - base)
  auto gather = [&](char* dst, size_t base, size_t o0, size_t o1) {
    for (size_t o = o0; o < o1;) {
      const size_t patch = o / pb, within = o - patch * pb;
      const size_t len = std::min(pb - within, o1 - o);
      copy_to_staging(dst + (o - base), source(patch) + within, len, streaming);
      o += len;
    }
    if (streaming) streaming_fence();
  };
  std::unique_ptr<GatherPool> pool;
  if (n_thr > 1) pool.reset(new GatherPool(n_thr));
  bool used[2] = {false, false};
  int b = 0;
  for (size_t c0 = 0; c0 < total; c0 += stage, b ^= 1) {
    const size_t c1 = std::min(total, c0 + stage), n = c1 - c0;
    if (used[b]) PXR_HIP(hipEventSynchronize(ctx->ev_stage[b]));            // the previous upload out of this buffer is done
    char* dst = static_cast<char*>(ctx->h_stage[b]);
    if (!pool || n < ((size_t)4 << 20)) {
      gather(dst, c0, c0, c1);
    } else {
      const std::function<void(unsigned)> job = [&](unsigned t) {
        // thread ranges on 4 KiB boundaries of the stream
        const size_t q = ((n + n_thr - 1) / n_thr + 4095) & ~(size_t)4095;
        const size_t o0 = std::min(n, q * t), o1 = std::min(n, q * (t + 1));
        if (o1 > o0) gather(dst, c0, c0 + o0, c0 + o1);
      };
      pool->run(job);
    }
    PXR_HIP(hipMemcpyAsync((char*)a->d_data + pb * (size_t)first + c0, dst, n, hipMemcpyHostToDevice, s));
    PXR_HIP(hipEventRecord(ctx->ev_stage[b], s));
    used[b] = true;
  }
  return PXR_OK;
}

static int upload_meta(pxr_arena* a, int64_t first, int64_t count, const int32_t* h_corners, const double* h_scales) {
  hipStream_t s = a->ctx->stream;
  if (h_corners)
    PXR_HIP(hipMemcpyAsync(a->d_corners + 2 * first, h_corners, sizeof(int32_t) * 2 * count, hipMemcpyHostToDevice, s));
  if (h_scales)
    PXR_HIP(hipMemcpyAsync(a->d_scales + 2 * first, h_scales, sizeof(double) * 2 * count, hipMemcpyHostToDevice, s));
  PXR_HIP(hipStreamSynchronize(s));
  return PXR_OK;
}

int pxr_arena_upload(pxr_arena* a, int64_t first, int64_t count, const void* h_patches,
                     const int32_t* h_corners, const double* h_scales) {
  PXR_REQUIRE(a, "pxr_arena_upload: arena is NULL");
  PXR_REQUIRE(first >= 0 && count >= 0 && first + count <= a->n,
              "pxr_arena_upload: range [%lld, %lld) outside arena of %lld patches", (long long)first,
              (long long)(first + count), (long long)a->n);
  if (count == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(a->ctx->device));
  if (h_patches) {
    if (int rc = upload_staged(a, first, count, h_patches, nullptr)) return rc;
  }
  return upload_meta(a, first, count, h_corners, h_scales);
}

int pxr_arena_upload_gather(pxr_arena* a, int64_t first, int64_t count, const void* const* h_patch_ptrs,
                            const int32_t* h_corners, const double* h_scales) {
  PXR_REQUIRE(a && h_patch_ptrs, "pxr_arena_upload_gather: NULL argument");
  PXR_REQUIRE(first >= 0 && count >= 0 && first + count <= a->n,
              "pxr_arena_upload_gather: range [%lld, %lld) outside arena of %lld patches", (long long)first,
              (long long)(first + count), (long long)a->n);
  if (count == 0) return PXR_OK;
  for (int64_t i = 0; i < count; ++i) PXR_REQUIRE(h_patch_ptrs[i], "pxr_arena_upload_gather: patch %lld is NULL", (long long)i);
  PXR_HIP(hipSetDevice(a->ctx->device));
  if (int rc = upload_staged(a, first, count, nullptr, h_patch_ptrs)) return rc;
  return upload_meta(a, first, count, h_corners, h_scales);
}

int pxr_arena_set_upsampling(pxr_arena* a, double upsampling_factor) {
  PXR_REQUIRE(a, "pxr_arena_set_upsampling: arena is NULL");
  PXR_REQUIRE(upsampling_factor > 0.0, "pxr_arena_set_upsampling: the factor must be positive");
  a->up = upsampling_factor;
  return PXR_OK;
}
double pxr_arena_upsampling(pxr_arena* a) { return a ? a->up : 1.0; }
void* pxr_arena_data(pxr_arena* a) { return a ? a->d_data : nullptr; }
int32_t* pxr_arena_corners(pxr_arena* a) { return a ? a->d_corners : nullptr; }
double* pxr_arena_scales(pxr_arena* a) { return a ? a->d_scales : nullptr; }
int64_t pxr_arena_size(pxr_arena* a) { return a ? a->n : 0; }

int pxr_set_deterministic(pxr_ctx* ctx, int on) {
  PXR_REQUIRE(ctx, "pxr_set_deterministic: NULL context");
  ctx->deterministic = on != 0;
  return PXR_OK;
}
int pxr_get_deterministic(pxr_ctx* ctx) { return ctx && ctx->deterministic ? 1 : 0; }

int pxr_set_gram_cache(pxr_ctx* ctx, int on) {
  PXR_REQUIRE(ctx, "pxr_set_gram_cache: NULL context");
  ctx->gram_cache = on != 0;
  if (!on && ctx->d_gram) {       // give the storage back
    PXR_HIP(hipSetDevice(ctx->device));
    PXR_HIP(hipStreamSynchronize(ctx->stream));
    PXR_HIP(hipFree(ctx->d_gram));
    ctx->d_gram = nullptr; ctx->gram_bytes = 0;
  }
  return PXR_OK;
}
int pxr_get_gram_cache(pxr_ctx* ctx) { return ctx && ctx->gram_cache ? 1 : 0; }

int pxr_set_iteration_callback(pxr_ctx* ctx, pxr_iteration_callback fn, void* user) {
  PXR_REQUIRE(ctx, "pxr_set_iteration_callback: NULL context");
  ctx->iter_cb = fn;
  ctx->iter_user = fn ? user : nullptr;
  return PXR_OK;
}

}  // extern "C"

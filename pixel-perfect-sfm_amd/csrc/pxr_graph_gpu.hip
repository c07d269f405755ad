// pxr_graph_gpu.hip -- match-graph labelling on the GPU: ComputeTrackLabels / ComputeScoreLabels /
// ComputeRootLabels (pixsfm/base/src/graph.cc:126-256) for a flat graph that already lives in HBM, with exactly the
// results of the native host version and of the oracle's restatement (tests/test_graph_gpu.py, tests/test_graph_labelling.py:
// bit for bit; networkx' connected components on conflict-free graphs).  PARITY UNPINNED w.r.t. graph.cc itself.
//
// The reference walks ALL matches once in descending (similarity, src, dst) order, merging two tracks unless they share
// an image.  A merge only involves the two tracks it touches and only nodes the match graph connects can ever merge, so
// the connected components of the (unconstrained) match graph are independent sub-problems -- the decomposition the
// native host version (pxr_graph.cpp) spreads over the CPU cores; here:
//   1. components: min-label hooking over the edges + pointer jumping (edge- / node-parallel, a few rounds);
//   2. the edges are bucketed by component (histogram, scan, scatter);
//   3. ONE WAVEFRONT PER COMPONENT: rank-sorts its bucket with the reference's comparator (every lane ranks its
//      elements against the bucket; ties broken by edge index), then lane 0 replays the reference's sequential
//      union-find on it -- path compression, conflict test over the two tracks' member lists, union by track size
//      with the reference's tie rule -- thousands of components in flight;
//   4. track ids = rank of the root nodes in node order (scan), scores = per-node sums over the incident edges in
//      ascending edge index (the reference's summation order, hence bit-identical), roots = per-track maximum of
//      (score, node index) by atomic max on order-preserving keys.
// Edges must be in Graph order (grouped by ascending source node, graph.cc:131-137).  Scratch comes from the context's
// grow-only workspace.
#include <hip/hip_runtime.h>

#include "pxr_internal.h"

namespace pxr {
namespace {

inline unsigned gblk(int64_t n, int b = 256) { return (unsigned)((n + b - 1) / b); }

__global__ void g_iota(int n, int* __restrict__ a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = i;
}
__global__ void g_fill(int64_t n, int v, int* __restrict__ a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

// Graph order + index range check: flag bit 0 = an edge index out of range, bit 1 = sources not ascending
__global__ void g_check(int64_t m, int n, const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int* __restrict__ flag) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  if (src[e] < 0 || src[e] >= n || dst[e] < 0 || dst[e] >= n) atomicOr(flag, 1);
  if (e + 1 < m && src[e] > src[e + 1]) atomicOr(flag, 2);
}

// components: hook the larger label under the smaller one, then compress
__global__ void g_hook(int64_t m, const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int* __restrict__ comp,
                       int* __restrict__ changed) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  const int a = comp[src[e]], b = comp[dst[e]];
  if (a == b) return;
  atomicMin(comp + (a > b ? a : b), a > b ? b : a);
  *changed = 1;
}
__global__ void g_jump(int n, int* __restrict__ comp, int* __restrict__ changed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int c = comp[i];
  while (comp[c] != c) c = comp[c];
  if (c != comp[i]) { comp[i] = c; *changed = 1; }
}

__global__ void g_count(int64_t m, const int64_t* __restrict__ key_node, const int* __restrict__ map /* or NULL */,
                        int* __restrict__ cnt) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  const int k = map ? map[key_node[e]] : (int)key_node[e];
  atomicAdd(cnt + k, 1);
}
__global__ void g_scatter(int64_t m, const int64_t* __restrict__ key_node, const int* __restrict__ map, int* __restrict__ cursor,
                          int* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  const int k = map ? map[key_node[e]] : (int)key_node[e];
  out[atomicAdd(cursor + k, 1)] = (int)e;
}

constexpr int kMaxComponentMatches = 32768;   // ~0.3 s of one wavefront; beyond it the host is faster
__global__ void g_max(int n, const int* __restrict__ cnt, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && cnt[i] > 0) atomicMax(out, cnt[i]);
}

// exclusive scan of a[0..n) in place, a[n] = total: one workgroup, every thread a contiguous chunk
__global__ __launch_bounds__(1024) void g_scan(int n, int* __restrict__ a) {
  __shared__ int part[1024];
  const int per = (n + 1023) / 1024, t0 = threadIdx.x * per, t1 = min(n, t0 + per);
  int s = 0;
  for (int i = t0; i < t1; ++i) s += a[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; } a[n] = run; }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int i = t0; i < t1; ++i) { const int v = a[i]; a[i] = run; run += v; }
}

struct EdgeKey { double sim; int64_t s, d; int e; };
// the reference sorts ascending tuples (sim, src, dst) and walks them backwards (graph.cc:145-146): "a before b" in that walk
__device__ __forceinline__ bool walks_before(const EdgeKey& a, const EdgeKey& b) {
  if (a.sim != b.sim) return a.sim > b.sim;
  if (a.s != b.s) return a.s > b.s;
  if (a.d != b.d) return a.d > b.d;
  return a.e > b.e;           // identical matches: any fixed order
}

__device__ int find_root(int i, int* __restrict__ parent) {
  int r = i;
  while (parent[r] != -1) r = parent[r];
  while (parent[i] != -1) { const int nx = parent[i]; parent[i] = r; i = nx; }   // path compression (graph.cc:116-124)
  return r;
}

// one wavefront per component (= per node that is a component's smallest node and has edges)
__global__ __launch_bounds__(256) void g_component(int n, const int* __restrict__ comp, const int* __restrict__ ptr,
                                                   const int* __restrict__ bucket, int* __restrict__ sorted,
                                                   const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                                   const double* __restrict__ sim, const int32_t* __restrict__ node_image,
                                                   int* __restrict__ parent, int* __restrict__ next, int* __restrict__ tail,
                                                   int* __restrict__ tsize) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= n || comp[c] != c) return;
  const int e0 = ptr[c], mcnt = ptr[c + 1] - e0;
  if (mcnt == 0) return;
  // rank sort: position of an edge in the reference's walk = number of edges walked before it
  for (int k = lane; k < mcnt; k += 64) {
    EdgeKey me;
    me.e = bucket[e0 + k]; me.sim = sim[me.e]; me.s = src[me.e]; me.d = dst[me.e];
    int rank = 0;
    for (int j = 0; j < mcnt; ++j) {
      EdgeKey o;
      o.e = bucket[e0 + j]; o.sim = sim[o.e]; o.s = src[o.e]; o.d = dst[o.e];
      rank += (o.e != me.e && walks_before(o, me)) ? 1 : 0;
    }
    sorted[e0 + rank] = me.e;
  }
  __threadfence();
  __builtin_amdgcn_wave_barrier();
  if (lane != 0) return;
  for (int k = 0; k < mcnt; ++k) {                       // the reference's sequential pass (graph.cc:156-184)
    const int e = sorted[e0 + k];
    const int r1 = find_root((int)src[e], parent), r2 = find_root((int)dst[e], parent);
    if (r1 == r2) continue;
    bool shared = false;                                 // do the two tracks share an image?
    for (int x = r1; x != -1 && !shared; x = next[x]) {
      const int ix = node_image[x];
      for (int y = r2; y != -1; y = next[y])
        if (node_image[y] == ix) { shared = true; break; }
    }
    if (shared) continue;
    // union by number of images in the track; on a tie root1 stays the root (graph.cc:172-182)
    const int big = tsize[r1] < tsize[r2] ? r2 : r1, small = tsize[r1] < tsize[r2] ? r1 : r2;
    parent[small] = big;
    next[tail[big]] = small; tail[big] = tail[small];
    tsize[big] += tsize[small];
  }
}

__global__ void g_root_flags(int n, const int* __restrict__ parent, int* __restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = parent[i] == -1 ? 1 : 0;
}
__global__ void g_assign(int n, const int* __restrict__ parent, const int* __restrict__ rootlabel, int64_t* __restrict__ labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = i;
  while (parent[r] != -1) r = parent[r];
  labels[i] = rootlabel[r];
}

// per node: its in-edges sorted by edge index (small lists), then the reference's summation order
__global__ void g_scores(int n, const int* __restrict__ optr, const int* __restrict__ iptr, int* __restrict__ ilist,
                         const int64_t* __restrict__ src, const int64_t* __restrict__ dst, const double* __restrict__ sim,
                         const int64_t* __restrict__ labels, double* __restrict__ scores) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int* li = ilist + iptr[i];
  const int ni = iptr[i + 1] - iptr[i];
  for (int a = 1; a < ni; ++a) {                         // insertion sort by edge index
    const int v = li[a];
    int b = a - 1;
    while (b >= 0 && li[b] > v) { li[b + 1] = li[b]; --b; }
    li[b + 1] = v;
  }
  const int64_t lab = labels[i];
  double acc = 0.0;
  int a = optr[i], a1 = optr[i + 1], b = 0;
  while (a < a1 || b < ni) {                             // ascending edge index over out- and in-edges (graph.cc:212-221);
                                                         // a self-match sits in both lists and adds twice, like the reference
    int e;
    if (b >= ni || (a < a1 && a < li[b])) e = a++; else e = li[b++];
    const int64_t other = src[e] == i ? dst[e] : src[e];
    if (labels[other] == lab) acc += sim[e];
  }
  scores[i] = acc;
}

__device__ __forceinline__ unsigned long long order_key(double v) {   // monotone map double -> uint64
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ void g_best_score(int n, const int64_t* __restrict__ labels, const double* __restrict__ scores,
                             unsigned long long* __restrict__ best) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicMax(best + labels[i], order_key(scores[i]));
}
__global__ void g_best_index(int n, const int64_t* __restrict__ labels, const double* __restrict__ scores,
                             const unsigned long long* __restrict__ best, int* __restrict__ best_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && order_key(scores[i]) == best[labels[i]]) atomicMax(best_idx + labels[i], i);
}
__global__ void g_mark_roots(int n, const int64_t* __restrict__ labels, const int* __restrict__ best_idx, uint8_t* __restrict__ is_root) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) is_root[i] = best_idx[labels[i]] == i ? 1 : 0;
}

}  // namespace
}  // namespace pxr

extern "C" int pxr_graph_labels_device(pxr_ctx* ctx, int64_t n_nodes, const int32_t* d_node_image, int64_t n_edges,
                                       const int64_t* d_edge_src, const int64_t* d_edge_dst, const double* d_edge_sim,
                                       int64_t* d_track_labels, double* d_scores, uint8_t* d_is_root,
                                       int64_t* h_n_tracks) {
  using namespace pxr;
  PXR_REQUIRE(ctx && n_nodes >= 0 && n_edges >= 0 && (n_nodes == 0 || (d_node_image && d_track_labels)) &&
                  (n_edges == 0 || (d_edge_src && d_edge_dst && d_edge_sim)),
              "pxr_graph_labels_device: NULL argument");
  PXR_REQUIRE(n_nodes < ((int64_t)1 << 31) - 2 && n_edges < ((int64_t)1 << 31) - 2, "pxr_graph_labels_device: more than 2^31 nodes or matches");
  PXR_REQUIRE(!d_is_root || d_scores, "pxr_graph_labels_device: the root labels need the score labels");
  if (h_n_tracks) *h_n_tracks = 0;
  if (n_nodes == 0) return PXR_OK;
  PXR_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int n = (int)n_nodes;
  const int64_t m = n_edges;
  // scratch (ints): comp n | ptr n+1 | cursor n+1 | bucket m | sorted m | parent n | next n | tail n | tsize n | flag 2
  //                 optr n+1 | iptr n+1 | ilist m | best_idx n ; + best keys (u64) n
  size_t off = 0;
  auto carve = [&](size_t count, size_t elem) { const size_t o = off; off += (count * elem + 255) & ~(size_t)255; return o; };
  const size_t o_comp = carve(n, 4), o_ptr = carve(n + 1, 4), o_cur = carve(n + 1, 4), o_bucket = carve(m, 4), o_sorted = carve(m, 4);
  const size_t o_parent = carve(n, 4), o_next = carve(n, 4), o_tail = carve(n, 4), o_tsize = carve(n, 4), o_flag = carve(4, 4);
  const size_t o_optr = carve(n + 1, 4), o_iptr = carve(n + 1, 4), o_ilist = carve(m, 4), o_bidx = carve(n, 4), o_best = carve(n, 8);
  if (off > ctx->workspace_bytes) {
    PXR_HIP(hipStreamSynchronize(st));
    if (ctx->d_workspace) { PXR_HIP(hipFree(ctx->d_workspace)); ctx->d_workspace = nullptr; ctx->workspace_bytes = 0; }
    PXR_HIP(hipMalloc(&ctx->d_workspace, off));
    ctx->workspace_bytes = off;
  }
  char* ws = static_cast<char*>(ctx->d_workspace);
  int* comp = (int*)(ws + o_comp); int* ptr = (int*)(ws + o_ptr); int* cursor = (int*)(ws + o_cur);
  int* bucket = (int*)(ws + o_bucket); int* sorted = (int*)(ws + o_sorted);
  int* parent = (int*)(ws + o_parent); int* next = (int*)(ws + o_next); int* tail = (int*)(ws + o_tail); int* tsize = (int*)(ws + o_tsize);
  int* flag = (int*)(ws + o_flag);
  int* optr = (int*)(ws + o_optr); int* iptr = (int*)(ws + o_iptr); int* ilist = (int*)(ws + o_ilist); int* best_idx = (int*)(ws + o_bidx);
  unsigned long long* best = (unsigned long long*)(ws + o_best);

  int h_flag[2] = {0, 0};
  PXR_HIP(hipMemsetAsync(flag, 0, 16, st));
  if (m > 0) hipLaunchKernelGGL(g_check, dim3(gblk(m)), dim3(256), 0, st, m, n, d_edge_src, d_edge_dst, flag);
  PXR_HIP(hipMemcpyAsync(h_flag, flag, 4, hipMemcpyDeviceToHost, st));
  PXR_HIP(hipStreamSynchronize(st));
  PXR_REQUIRE(!(h_flag[0] & 1), "pxr_graph_labels_device: a match references a node out of range");
  PXR_REQUIRE(!(h_flag[0] & 2), "pxr_graph_labels_device: matches must be in Graph order (grouped by ascending source node)");
  // 1. components
  hipLaunchKernelGGL(g_iota, dim3(gblk(n)), dim3(256), 0, st, n, comp);
  for (int64_t round = 0; m > 0; ++round) {             // until no label moves (hook + full pointer jumping: O(log n) rounds)
    PXR_REQUIRE(round <= (int64_t)n + 1, "pxr_graph_labels_device: the component labelling did not converge");
    PXR_HIP(hipMemsetAsync(flag + 1, 0, 4, st));
    hipLaunchKernelGGL(g_hook, dim3(gblk(m)), dim3(256), 0, st, m, d_edge_src, d_edge_dst, comp, flag + 1);
    hipLaunchKernelGGL(g_jump, dim3(gblk(n)), dim3(256), 0, st, n, comp, flag + 1);
    PXR_HIP(hipMemcpyAsync(h_flag + 1, flag + 1, 4, hipMemcpyDeviceToHost, st));
    PXR_HIP(hipStreamSynchronize(st));
    if (!h_flag[1]) break;
  }
  // 2. buckets of edges per component (keyed by the component of the source node)
  PXR_HIP(hipMemsetAsync(ptr, 0, sizeof(int) * ((size_t)n + 1), st));
  if (m > 0) hipLaunchKernelGGL(g_count, dim3(gblk(m)), dim3(256), 0, st, m, d_edge_src, (const int*)comp, ptr);
  if (m > 0) {
    // g_component orders a component's matches with an O(m_c^2) rank sort in ONE wavefront: fine for tracks-sized
    // components, quadratic when bad matches chain a scene into one giant component -- refuse those (the caller
    // falls back to the host labelling, pxr_graph_track_labels) instead of running a single wave for minutes
    hipLaunchKernelGGL(g_max, dim3(gblk(n)), dim3(256), 0, st, n, (const int*)ptr, flag + 2);
    PXR_HIP(hipMemcpyAsync(h_flag, flag + 2, 4, hipMemcpyDeviceToHost, st));
    PXR_HIP(hipStreamSynchronize(st));
    if (h_flag[0] > kMaxComponentMatches)
      return set_error(PXR_EUNSUPPORTED, "pxr_graph_labels_device: a connected component holds %d matches (limit %d): label on the host",
                       h_flag[0], kMaxComponentMatches);
  }
  hipLaunchKernelGGL(g_scan, dim3(1), dim3(1024), 0, st, n, ptr);
  PXR_HIP(hipMemcpyAsync(cursor, ptr, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToDevice, st));
  if (m > 0) hipLaunchKernelGGL(g_scatter, dim3(gblk(m)), dim3(256), 0, st, m, d_edge_src, (const int*)comp, cursor, bucket);
  // 3. constrained merges, one wavefront per component
  hipLaunchKernelGGL(g_fill, dim3(gblk(n)), dim3(256), 0, st, (int64_t)n, -1, parent);
  hipLaunchKernelGGL(g_fill, dim3(gblk(n)), dim3(256), 0, st, (int64_t)n, -1, next);
  hipLaunchKernelGGL(g_iota, dim3(gblk(n)), dim3(256), 0, st, n, tail);
  hipLaunchKernelGGL(g_fill, dim3(gblk(n)), dim3(256), 0, st, (int64_t)n, 1, tsize);
  if (m > 0)
    hipLaunchKernelGGL(g_component, dim3(gblk(n, 4)), dim3(256), 0, st, n, (const int*)comp, (const int*)ptr, (const int*)bucket, sorted,
                       d_edge_src, d_edge_dst, d_edge_sim, d_node_image, parent, next, tail, tsize);
  // 4. track ids in node order
  hipLaunchKernelGGL(g_root_flags, dim3(gblk(n)), dim3(256), 0, st, n, (const int*)parent, cursor);
  hipLaunchKernelGGL(g_scan, dim3(1), dim3(1024), 0, st, n, cursor);
  hipLaunchKernelGGL(g_assign, dim3(gblk(n)), dim3(256), 0, st, n, (const int*)parent, (const int*)cursor, d_track_labels);
  int h_tracks = 0;
  PXR_HIP(hipMemcpyAsync(&h_tracks, cursor + n, 4, hipMemcpyDeviceToHost, st));
  if (d_scores) {
    PXR_HIP(hipMemsetAsync(optr, 0, sizeof(int) * ((size_t)n + 1), st));
    PXR_HIP(hipMemsetAsync(iptr, 0, sizeof(int) * ((size_t)n + 1), st));
    if (m > 0) {
      hipLaunchKernelGGL(g_count, dim3(gblk(m)), dim3(256), 0, st, m, d_edge_src, (const int*)nullptr, optr);
      hipLaunchKernelGGL(g_count, dim3(gblk(m)), dim3(256), 0, st, m, d_edge_dst, (const int*)nullptr, iptr);
    }
    hipLaunchKernelGGL(g_scan, dim3(1), dim3(1024), 0, st, n, optr);
    hipLaunchKernelGGL(g_scan, dim3(1), dim3(1024), 0, st, n, iptr);
    PXR_HIP(hipMemcpyAsync(comp, iptr, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, st));     // comp is free now: cursor of the in-lists
    if (m > 0) hipLaunchKernelGGL(g_scatter, dim3(gblk(m)), dim3(256), 0, st, m, d_edge_dst, (const int*)nullptr, comp, ilist);
    hipLaunchKernelGGL(g_scores, dim3(gblk(n)), dim3(256), 0, st, n, (const int*)optr, (const int*)iptr, ilist, d_edge_src, d_edge_dst,
                       d_edge_sim, (const int64_t*)d_track_labels, d_scores);
  }
  if (d_is_root) {
    PXR_HIP(hipMemsetAsync(best, 0, sizeof(unsigned long long) * (size_t)n, st));
    hipLaunchKernelGGL(g_fill, dim3(gblk(n)), dim3(256), 0, st, (int64_t)n, -1, best_idx);
    hipLaunchKernelGGL(g_best_score, dim3(gblk(n)), dim3(256), 0, st, n, (const int64_t*)d_track_labels, (const double*)d_scores, best);
    hipLaunchKernelGGL(g_best_index, dim3(gblk(n)), dim3(256), 0, st, n, (const int64_t*)d_track_labels, (const double*)d_scores,
                       (const unsigned long long*)best, best_idx);
    hipLaunchKernelGGL(g_mark_roots, dim3(gblk(n)), dim3(256), 0, st, n, (const int64_t*)d_track_labels, (const int*)best_idx, d_is_root);
  }
  PXR_HIP(hipGetLastError());
  PXR_HIP(hipStreamSynchronize(st));
  if (h_n_tracks) *h_n_tracks = h_tracks;
  return PXR_OK;
}

// pxr_graph.cpp -- track / score / root labelling of the match graph (host code, native like the
// reference's).
//
// Reference: ComputeTrackLabels / ComputeScoreLabels / ComputeRootLabels (pixsfm/base/src/graph.cc:126-256),
// called from KeypointAdjuster.refine (keypoint_adjustment/main.py:111-118) right before the optimisers.
// The track labelling is a maximum-spanning-forest union-find over the matches in descending
// (similarity, src, dst) order that never merges two components sharing an image.  Inside one connected
// component of the match graph every merge depends on the earlier ones (sequential), but the components are
// independent, which is where the parallelism is: they are solved concurrently on the host cores (SURVEY 8f
// row 3: once the KA solve takes milliseconds, neither Python nor one core will do).  Flat arrays in / out.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <thread>
#include <tuple>
#include <vector>

#include "pxr_internal.h"

namespace {

int64_t find_root(int64_t i, std::vector<int64_t>& parent) {
  int64_t r = i;
  while (parent[r] != -1) r = parent[r];
  while (parent[i] != -1) {   // path compression (graph.cc:116-124)
    const int64_t nx = parent[i];
    parent[i] = r;
    i = nx;
  }
  return r;
}

}  // namespace

extern "C" int pxr_graph_track_labels(int64_t n_nodes, const int32_t* node_image, int64_t n_edges,
                                      const int64_t* edge_src, const int64_t* edge_dst, const double* edge_sim,
                                      int64_t* track_labels, int64_t* n_tracks_out) {
  PXR_REQUIRE(n_nodes >= 0 && n_edges >= 0 && (n_nodes == 0 || (node_image && track_labels)) &&
                  (n_edges == 0 || (edge_src && edge_dst && edge_sim)),
              "pxr_graph_track_labels: NULL argument");
  for (int64_t e = 0; e < n_edges; ++e)
    PXR_REQUIRE(edge_src[e] >= 0 && edge_src[e] < n_nodes && edge_dst[e] >= 0 && edge_dst[e] < n_nodes,
                "pxr_graph_track_labels: edge %lld out of range", (long long)e);
  // The reference walks ALL matches in one descending order.  A merge decision only involves the two components
  // it touches, and two nodes can only ever be merged if the match graph connects them, so the connected components
  // of the (unconstrained) match graph are independent sub-problems: bucket the matches by component, then sort and
  // merge every bucket on its own -- same comparator, hence the same relative order and the same result -- on all
  // host cores.  (At BASELINE configs[1], 450k matches, the single global pass took 10x the GPU KA solve.)
  const auto t0 = std::chrono::steady_clock::now();
  const bool verbose = std::getenv("PXR_VERBOSE") != nullptr;
  auto mark = [&](const char* what) {
    if (verbose)
      fprintf(stderr, "[pxr_graph_track_labels] %-26s at %.2f ms\n", what,
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  };
  std::vector<int64_t> comp((size_t)n_nodes, -1);
  for (int64_t e = 0; e < n_edges; ++e) {
    const int64_t a = find_root(edge_src[e], comp), b = find_root(edge_dst[e], comp);
    if (a != b) comp[(size_t)std::max(a, b)] = std::min(a, b);
  }
  std::vector<int64_t> bucket_ptr((size_t)n_nodes + 1, 0);
  std::vector<int64_t> edge_comp((size_t)n_edges);
  for (int64_t e = 0; e < n_edges; ++e) {
    edge_comp[(size_t)e] = find_root(edge_src[e], comp);
    ++bucket_ptr[(size_t)edge_comp[(size_t)e] + 1];
  }
  for (int64_t i = 0; i < n_nodes; ++i) bucket_ptr[(size_t)i + 1] += bucket_ptr[(size_t)i];
  std::vector<int64_t> bucket((size_t)n_edges);
  {
    std::vector<int64_t> fill(bucket_ptr.begin(), bucket_ptr.end() - 1);
    for (int64_t e = 0; e < n_edges; ++e) bucket[(size_t)fill[(size_t)edge_comp[(size_t)e]]++] = e;
  }
  std::vector<int64_t> comps;                       // components that have matches
  for (int64_t i = 0; i < n_nodes; ++i)
    if (bucket_ptr[(size_t)i + 1] > bucket_ptr[(size_t)i]) comps.push_back(i);

  mark("components + buckets");
  typedef std::tuple<double, int64_t, int64_t> edge_t;
  std::vector<int64_t> parent((size_t)n_nodes, -1);
  std::vector<std::vector<int32_t>> images((size_t)n_nodes);   // sorted image ids of each root's component
  auto solve_component = [&](int64_t c, std::vector<edge_t>& edges, std::vector<int32_t>& merged) {
    edges.clear();
    for (int64_t k = bucket_ptr[(size_t)c]; k < bucket_ptr[(size_t)c + 1]; ++k) {
      const int64_t e = bucket[(size_t)k];
      edges.push_back(std::make_tuple(edge_sim[e], edge_src[e], edge_dst[e]));
    }
    std::sort(edges.begin(), edges.end());          // ascending tuples, then walked backwards (graph.cc:145-146)
    for (auto it = edges.rbegin(); it != edges.rend(); ++it) {
      const int64_t r1 = find_root(std::get<1>(*it), parent), r2 = find_root(std::get<2>(*it), parent);
      if (r1 == r2) continue;
      std::vector<int32_t>& a = images[(size_t)r1];
      std::vector<int32_t>& b = images[(size_t)r2];
      if (a.empty()) a.push_back(node_image[r1]);   // singleton sets are materialised on first use
      if (b.empty()) b.push_back(node_image[r2]);
      bool shared = false;                          // std::set_intersection non-empty (graph.cc:163-170)
      for (size_t x = 0, y = 0; x < a.size() && y < b.size();) {
        if (a[x] == b[y]) { shared = true; break; }
        if (a[x] < b[y]) ++x; else ++y;
      }
      if (shared) continue;
      merged.resize(a.size() + b.size());
      std::merge(a.begin(), a.end(), b.begin(), b.end(), merged.begin());
      if (a.size() < b.size()) {                    // union by component size (graph.cc:172-182)
        parent[(size_t)r1] = r2;
        b = merged; std::vector<int32_t>().swap(a);
      } else {
        parent[(size_t)r2] = r1;
        a = merged; std::vector<int32_t>().swap(b);
      }
    }
  };
  unsigned n_threads = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
  if (const char* env = std::getenv("PXR_GRAPH_THREADS")) n_threads = (unsigned)std::max(1, atoi(env));
  if (n_edges < 20000 || comps.size() < 2 * (size_t)n_threads) n_threads = 1;
  if (n_threads == 1) {
    std::vector<edge_t> edges;
    std::vector<int32_t> merged;
    for (int64_t c : comps) solve_component(c, edges, merged);
  } else {
    // components are disjoint in nodes, so the workers never touch the same parent[] / images[] entries; work is dealt
    // in blocks of components through one atomic counter (giant components do not stall a static partition)
    std::atomic<size_t> next(0);
    const size_t block = 64;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < n_threads; ++t)
      pool.emplace_back([&]() {
        std::vector<edge_t> edges;
        std::vector<int32_t> merged;
        for (;;) {
          const size_t b0 = next.fetch_add(block);
          if (b0 >= comps.size()) break;
          for (size_t k = b0; k < std::min(comps.size(), b0 + block); ++k) solve_component(comps[k], edges, merged);
        }
      });
    for (std::thread& th : pool) th.join();
  }
  mark("constrained merges");
  int64_t n_tracks = 0;
  for (int64_t i = 0; i < n_nodes; ++i) track_labels[i] = parent[(size_t)i] == -1 ? n_tracks++ : -1;
  for (int64_t i = 0; i < n_nodes; ++i)
    if (track_labels[i] == -1) track_labels[i] = track_labels[find_root(i, parent)];
  if (n_tracks_out) *n_tracks_out = n_tracks;
  return PXR_OK;
}

extern "C" int pxr_graph_score_labels(int64_t n_nodes, int64_t n_edges, const int64_t* edge_src,
                                      const int64_t* edge_dst, const double* edge_sim,
                                      const int64_t* track_labels, double* scores) {
  PXR_REQUIRE(n_nodes >= 0 && n_edges >= 0 && (n_nodes == 0 || (track_labels && scores)) &&
                  (n_edges == 0 || (edge_src && edge_dst && edge_sim)),
              "pxr_graph_score_labels: NULL argument");
  std::fill(scores, scores + n_nodes, 0.0);
  for (int64_t e = 0; e < n_edges; ++e) {          // edge order = node order x out_matches order (graph.cc:212-221)
    const int64_t s = edge_src[e], d = edge_dst[e];
    PXR_REQUIRE(s >= 0 && s < n_nodes && d >= 0 && d < n_nodes, "pxr_graph_score_labels: edge %lld out of range", (long long)e);
    if (track_labels[s] == track_labels[d]) { scores[s] += edge_sim[e]; scores[d] += edge_sim[e]; }
  }
  return PXR_OK;
}

extern "C" int pxr_graph_root_labels(int64_t n_nodes, const int64_t* track_labels, const double* scores,
                                     uint8_t* is_root) {
  PXR_REQUIRE(n_nodes >= 0 && (n_nodes == 0 || (track_labels && scores && is_root)), "pxr_graph_root_labels: NULL argument");
  // The reference sorts all nodes by descending (score, index) and takes the first node of every track
  // (graph.cc:225-256): that is the per-track maximum of the pair, one linear pass.
  int64_t n_tracks = 0;
  for (int64_t i = 0; i < n_nodes; ++i) {
    PXR_REQUIRE(track_labels[i] >= 0, "pxr_graph_root_labels: negative track label");
    n_tracks = std::max(n_tracks, track_labels[i] + 1);
  }
  std::vector<int64_t> best((size_t)n_tracks, -1);
  for (int64_t i = 0; i < n_nodes; ++i) {
    int64_t& b = best[(size_t)track_labels[i]];
    if (b < 0 || std::make_pair(scores[i], i) > std::make_pair(scores[b], b)) b = i;
  }
  std::fill(is_root, is_root + n_nodes, (uint8_t)0);
  for (int64_t t = 0; t < n_tracks; ++t)
    if (best[(size_t)t] >= 0) is_root[best[(size_t)t]] = 1;
  return PXR_OK;
}

// TopologicalKeypointOptimizer::SetUp + FeatureMetricKeypointOptimizer::AddIntraResiduals
// (keypoint_adjustment/src/topological_keypoint_optimizer.h:97-175, featuremetric_keypoint_optimizer.h:158-202):
// which matches become residual blocks, with which ScaledLoss weight, in the reference's insertion order.
//   - intra-track matches only (:111-116); a match between two aliases of one keypoint is skipped (:147-150);
//   - root_edges_only keeps blocks with a root endpoint (featuremetric_keypoint_optimizer.h:169-172);
//   - root_regularize_weight > 0 adds one block node -> track root for nodes no match connects to their
//     root (:156-170; an unknown root defaults to node 0 like the reference's unordered_map::operator[]).
// nodes_in_problem: NULL (all nodes in order) or the n_in_problem node indices to enumerate.  The edge arrays are
// in Graph order (grouped by ascending source node).  out_*: capacity 3 * n_edges; *n_out receives the count.
extern "C" int pxr_ka_build_edges(int64_t n_nodes, const int32_t* node_image, const int32_t* node_feature,
                                  int64_t n_edges, const int64_t* edge_src, const int64_t* edge_dst,
                                  const double* edge_sim, const int64_t* track_labels, const uint8_t* root_labels,
                                  const int64_t* nodes_in_problem, int64_t n_in_problem, int weight_by_sim,
                                  int root_edges_only, double root_regularize_weight, int64_t* out_src,
                                  int64_t* out_dst, double* out_w, int64_t* n_out) {
  PXR_REQUIRE(n_nodes >= 0 && n_edges >= 0 && n_out && (n_nodes == 0 || (node_image && node_feature && track_labels && root_labels)) &&
                  (n_edges == 0 || (edge_src && edge_dst && edge_sim && out_src && out_dst && out_w)),
              "pxr_ka_build_edges: NULL argument");
  std::vector<int64_t> row_ptr((size_t)n_nodes + 1, 0);
  for (int64_t e = 0; e < n_edges; ++e) {
    PXR_REQUIRE(edge_src[e] >= 0 && edge_src[e] < n_nodes && edge_dst[e] >= 0 && edge_dst[e] < n_nodes,
                "pxr_ka_build_edges: edge %lld out of range", (long long)e);
    PXR_REQUIRE(e == 0 || edge_src[e] >= edge_src[e - 1], "pxr_ka_build_edges: edges must be grouped by ascending source node");
    ++row_ptr[(size_t)edge_src[e] + 1];
  }
  for (int64_t i = 0; i < n_nodes; ++i) row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
  const bool regularize = root_regularize_weight > 0.0;
  std::vector<char> connected((size_t)n_nodes, 0);
  std::vector<int64_t> track_root;   // by track label, default 0
  if (regularize) {
    int64_t n_tracks = 0;
    for (int64_t i = 0; i < n_nodes; ++i) n_tracks = std::max(n_tracks, track_labels[i] + 1);
    track_root.assign((size_t)n_tracks, 0);
  }
  std::vector<int64_t> cand;         // edge ids of the intra-track matches of the problem, in enumeration order
  const int64_t n_enum = nodes_in_problem ? n_in_problem : n_nodes;
  for (int64_t x = 0; x < n_enum; ++x) {
    const int64_t i = nodes_in_problem ? nodes_in_problem[x] : x;
    PXR_REQUIRE(i >= 0 && i < n_nodes, "pxr_ka_build_edges: node %lld out of range", (long long)i);
    for (int64_t e = row_ptr[(size_t)i]; e < row_ptr[(size_t)i + 1]; ++e) {
      const int64_t j = edge_dst[e];
      if (track_labels[i] != track_labels[j]) continue;
      cand.push_back(e);
      if (regularize) {
        if (root_labels[i]) { track_root[(size_t)track_labels[i]] = i; connected[(size_t)i] = connected[(size_t)j] = 1; }
        if (root_labels[j]) { track_root[(size_t)track_labels[j]] = j; connected[(size_t)i] = connected[(size_t)j] = 1; }
      }
    }
  }
  int64_t m = 0;
  auto add = [&](int64_t a, int64_t b, double w) {   // AddIntraResiduals
    if (track_labels[a] != track_labels[b]) return;
    if (root_edges_only && !root_labels[a] && !root_labels[b]) return;
    out_src[m] = a; out_dst[m] = b; out_w[m] = w; ++m;
  };
  for (const int64_t e : cand) {
    const int64_t i = edge_src[e], j = edge_dst[e];
    if (node_image[i] == node_image[j] && node_feature[i] == node_feature[j]) continue;   // same keypoint
    add(i, j, weight_by_sim ? edge_sim[e] : 1.0);
    if (regularize) {
      const int64_t both[2] = {i, j};
      for (const int64_t k : both) {
        if (!connected[(size_t)k]) {
          add(k, track_root[(size_t)track_labels[k]], root_regularize_weight);
          connected[(size_t)k] = 1;
        }
      }
    }
  }
  *n_out = m;
  return PXR_OK;
}

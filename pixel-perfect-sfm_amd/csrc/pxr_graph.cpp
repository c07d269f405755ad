// pxr_graph.cpp -- track / score / root labelling of the match graph (host code, native like the
// reference's).
//
// Reference: ComputeTrackLabels / ComputeScoreLabels / ComputeRootLabels (pixsfm/base/src/graph.cc:126-256),
// called from KeypointAdjuster.refine (keypoint_adjustment/main.py:111-118) right before the optimisers.
// The track labelling is a maximum-spanning-forest union-find over the matches in descending
// (similarity, src, dst) order that never merges two components sharing an image: every merge depends
// on all earlier ones, so it stays a sequential host pass -- but once the KA solve takes
// milliseconds it must not be Python.  Flat arrays in, flat arrays out.
#include <algorithm>
#include <cstdint>
#include <numeric>
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
  typedef std::tuple<double, int64_t, int64_t> edge_t;
  std::vector<edge_t> edges((size_t)n_edges);
  for (int64_t e = 0; e < n_edges; ++e) {
    PXR_REQUIRE(edge_src[e] >= 0 && edge_src[e] < n_nodes && edge_dst[e] >= 0 && edge_dst[e] < n_nodes,
                "pxr_graph_track_labels: edge %lld out of range", (long long)e);
    edges[(size_t)e] = std::make_tuple(edge_sim[e], edge_src[e], edge_dst[e]);
  }
  std::sort(edges.begin(), edges.end());          // ascending tuples, then reversed (graph.cc:145-146)
  std::reverse(edges.begin(), edges.end());
  std::vector<int64_t> parent((size_t)n_nodes, -1);
  std::vector<std::vector<int32_t>> images((size_t)n_nodes);   // sorted image ids of each root's component
  for (int64_t i = 0; i < n_nodes; ++i) images[(size_t)i].push_back(node_image[i]);
  std::vector<int32_t> merged;
  for (const edge_t& ed : edges) {
    const int64_t r1 = find_root(std::get<1>(ed), parent), r2 = find_root(std::get<2>(ed), parent);
    if (r1 == r2) continue;
    const std::vector<int32_t>& a = images[(size_t)r1];
    const std::vector<int32_t>& b = images[(size_t)r2];
    bool shared = false;                           // std::set_intersection non-empty (graph.cc:163-170)
    for (size_t x = 0, y = 0; x < a.size() && y < b.size();) {
      if (a[x] == b[y]) { shared = true; break; }
      if (a[x] < b[y]) ++x; else ++y;
    }
    if (shared) continue;
    merged.resize(a.size() + b.size());
    std::merge(a.begin(), a.end(), b.begin(), b.end(), merged.begin());
    if (a.size() < b.size()) {                     // union by component size (graph.cc:172-182)
      parent[(size_t)r1] = r2;
      images[(size_t)r2] = merged; images[(size_t)r1].clear();
    } else {
      parent[(size_t)r2] = r1;
      images[(size_t)r1] = merged; images[(size_t)r2].clear();
    }
  }
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
  std::vector<std::pair<double, int64_t>> order((size_t)n_nodes);
  int64_t n_tracks = 0;
  for (int64_t i = 0; i < n_nodes; ++i) {
    PXR_REQUIRE(track_labels[i] >= 0, "pxr_graph_root_labels: negative track label");
    order[(size_t)i] = std::make_pair(scores[i], i);
    n_tracks = std::max(n_tracks, track_labels[i] + 1);
  }
  std::sort(order.begin(), order.end());           // ascending pairs, then reversed (graph.cc:238-239)
  std::reverse(order.begin(), order.end());
  std::vector<char> has_root((size_t)n_tracks, 0);
  std::fill(is_root, is_root + n_nodes, (uint8_t)0);
  for (const auto& it : order) {
    const int64_t i = it.second;
    if (has_root[(size_t)track_labels[i]]) continue;
    is_root[i] = 1;
    has_root[(size_t)track_labels[i]] = 1;
  }
  return PXR_OK;
}

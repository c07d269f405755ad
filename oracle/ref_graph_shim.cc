// ref_graph_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN match-graph code where it lies under /root/reference:
//   pixsfm/base/src/graph.h, graph.cc   (Graph::RegisterMatches, ComputeTrackLabels, ComputeScoreLabels,
//                                        ComputeRootLabels, CountTrackEdges)
// against the stub headers oracle/ref_stubs/{colmap/util/types.h, util/src/simple_logger.h,
// util/src/log_exceptions.h}.  Output: oracle/_ref/libpxo_ref_graph.so.  Nothing of the reference is copied
// into this repository; the C entry point below only feeds flat arrays through its public interface.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "base/src/graph.cc"

extern "C" {

// pairs: n_pairs x 2 image indices ("im<k>" names); matches of pair p: rows match_ptr[p] .. match_ptr[p+1] of
// matches (feature_idx1, feature_idx2) and sims.  Outputs sized by the caller with capacity max_nodes:
// node_image / node_feature (graph node order), track labels, scores, roots, edges per track (capacity max_nodes).
// Returns the number of nodes, or -1 if the capacity is too small.
int64_t pxo_ref_graph_labels(int64_t n_pairs, const int32_t* pairs, const int64_t* match_ptr, const int64_t* matches,
                             const double* sims, int64_t max_nodes, int32_t* node_image, int32_t* node_feature,
                             int64_t* track_labels, double* scores, uint8_t* roots, int64_t* n_tracks,
                             int64_t* track_edges) {
  pixsfm::Graph graph;
  for (int64_t p = 0; p < n_pairs; ++p) {
    const int64_t m0 = match_ptr[p], m1 = match_ptr[p + 1];
    std::vector<size_t> mm(2 * (m1 - m0));
    for (int64_t i = 0; i < 2 * (m1 - m0); ++i) mm[i] = (size_t)matches[2 * m0 + i];
    std::vector<double> ss(sims + m0, sims + m1);
    graph.RegisterMatches("im" + std::to_string(pairs[2 * p]), "im" + std::to_string(pairs[2 * p + 1]), mm.data(),
                          ss.data(), (size_t)(m1 - m0));
  }
  const int64_t n = (int64_t)graph.nodes.size();
  if (n > max_nodes) return -1;
  std::vector<size_t> labels = pixsfm::ComputeTrackLabels(graph);
  std::vector<double> sc = pixsfm::ComputeScoreLabels(graph, labels);
  std::vector<bool> rt = pixsfm::ComputeRootLabels(graph, labels, sc);
  std::vector<size_t> te = pixsfm::CountTrackEdges(graph, labels);
  for (int64_t i = 0; i < n; ++i) {
    // the image id the graph assigned (order of first appearance) -> the caller's image index through the name
    const std::string& name = graph.image_id_to_name.at(graph.nodes[i]->image_id);
    node_image[i] = std::stoi(name.substr(2));
    node_feature[i] = (int32_t)graph.nodes[i]->feature_idx;
    track_labels[i] = (int64_t)labels[i];
    scores[i] = sc[i];
    roots[i] = rt[i] ? 1 : 0;
  }
  *n_tracks = (int64_t)te.size();
  for (size_t t = 0; t < te.size() && (int64_t)t < max_nodes; ++t) track_edges[t] = (int64_t)te[t];
  return n;
}

}  // extern "C"

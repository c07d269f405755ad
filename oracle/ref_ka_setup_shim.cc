// ref_ka_setup_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN keypoint-adjustment problem construction where it lies under /root/reference and RECORDS
// what it hands to Ceres (nothing is solved):
//   pixsfm/keypoint_adjustment/src/topological_keypoint_optimizer.h    TopologicalKeypointOptimizer::Run / SetUp  (edge
//       enumeration over out_matches, the keypoint-alias skip, root regularisation blocks)
//   pixsfm/keypoint_adjustment/src/featuremetric_keypoint_optimizer.h  AddIntraResiduals (root_edges_only, ScaledLoss weight)
//   pixsfm/keypoint_adjustment/src/keypoint_optimizer.h                ParameterizeKeypoints (constant nodes, box bounds)
//   pixsfm/keypoint_adjustment/src/keypoint_adjustment_options.{h,cc}  KeypointAdjustmentSetup
//   pixsfm/base/src/graph.{h,cc}                                       Graph, labels
// against a recording ceres::Problem (oracle/ref_stubs/interp/ceres/ceres.h), in-memory stand-ins for the HDF5-backed
// feature containers (oracle/ref_stubs/kasetup/) and the stub headers of the other shims.
// Output: oracle/_ref/libpxo_ref_ka_setup.so.  Nothing of the reference is copied into this repository.
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

#include "base/src/graph.cc"
#include "keypoint_adjustment/src/keypoint_adjustment_options.cc"
#include "keypoint_adjustment/src/featuremetric_keypoint_optimizer.h"

namespace pixsfm {
template <typename dtype>
FeaturePatch<dtype>::FeaturePatch() : data_ptr_(nullptr) {}

struct MetaPatch : public FeaturePatch<half> {
  MetaPatch(int H, int W, int cx, int cy, double sx, double sy) {
    this->data_ptr_ = nullptr;
    this->shape_ = {H, W, 128};
    this->corner_[0] = cx; this->corner_[1] = cy;
    this->scale_[0] = sx; this->scale_[1] = sy;
  }
};

// access to the recording problem of a finished optimizer
struct Recorder : public FeatureMetricKeypointOptimizer {
  using FeatureMetricKeypointOptimizer::FeatureMetricKeypointOptimizer;
  ceres::Problem* problem() { return problem_.get(); }
};
}  // namespace pixsfm

extern "C" {

// The graph is registered pair by pair exactly as in ref_graph_shim.cc (image index k <-> name "im<k>").
// keypoints: per image k the rows kp_ptr[k] .. kp_ptr[k+1] of kp (x, y); patches: per NODE (graph order) corner (2 ints) and
// scale (2 doubles), all H x W.  alias_of: per image, -1 or the index of an image whose keypoint ARRAY it shares (the
// reference compares data pointers to avoid optimising a keypoint against itself).  nodes_in_problem: n_in node indices (NULL: all).
// const_images: n_const image indices (KeypointAdjustmentSetup::SetImageConstant); const_roots: also every root node constant.
// Outputs: blocks (capacity max_blocks): source / destination node and ScaledLoss weight in the order the reference added them;
// per node: constant flag, 4 bounds (lower x, lower y, upper x, upper y; NaN = not set), will-be-optimised via has_bounds|const.
// Returns the number of residual blocks, or -1 on overflow.
int64_t pxo_ref_ka_setup(int64_t n_pairs, const int32_t* pairs, const int64_t* match_ptr, const int64_t* matches, const double* sims,
                         int n_images, const int64_t* kp_ptr, const double* kp, const int32_t* alias_of, int H, int W,
                         const int32_t* node_corner, const double* node_scale, const int64_t* nodes_in_problem, int64_t n_in,
                         const int32_t* const_images, int n_const, int const_roots, int weight_by_sim, int root_edges_only,
                         double root_regularize_weight, double bound, int64_t max_blocks, int64_t* blk_src, int64_t* blk_dst,
                         double* blk_w, uint8_t* node_const, double* node_bounds) try {
  using namespace pixsfm;
  Graph graph;
  for (int64_t p = 0; p < n_pairs; ++p) {
    const int64_t m0 = match_ptr[p], m1 = match_ptr[p + 1];
    std::vector<size_t> mm(2 * (m1 - m0));
    for (int64_t i = 0; i < 2 * (m1 - m0); ++i) mm[i] = (size_t)matches[2 * m0 + i];
    std::vector<double> ss(sims + m0, sims + m1);
    graph.RegisterMatches("im" + std::to_string(pairs[2 * p]), "im" + std::to_string(pairs[2 * p + 1]), mm.data(), ss.data(),
                          (size_t)(m1 - m0));
  }
  const size_t n = graph.nodes.size();
  std::vector<size_t> labels = ComputeTrackLabels(graph);
  std::vector<double> scores = ComputeScoreLabels(graph, labels);
  std::vector<bool> roots = ComputeRootLabels(graph, labels, scores);

  // keypoint arrays: one matrix per image name
  MapNameKeypoints keypoints;
  for (int k = 0; k < n_images; ++k) {
    const int64_t r0 = kp_ptr[k], r1 = kp_ptr[k + 1];
    KeypointMatrixd m((int)(r1 - r0), 2);
    for (int64_t r = r0; r < r1; ++r) { m(r - r0, 0) = kp[2 * r]; m(r - r0, 1) = kp[2 * r + 1]; }
    keypoints["im" + std::to_string(k)] = m;
  }
  (void)alias_of;   // distinct arrays per image: the alias skip of SetUp (:148-151) cannot fire across images here

  // feature view: patch metadata per node
  FeatureView<half> fview;
  std::vector<std::unique_ptr<MetaPatch>> owned;
  for (size_t i = 0; i < n; ++i) {
    const FeatureNode* node = graph.nodes[i];
    owned.emplace_back(new MetaPatch(H, W, node_corner[2 * i], node_corner[2 * i + 1], node_scale[2 * i], node_scale[2 * i + 1]));
    fview.maps[node->image_id].patches[node->feature_idx] = owned.back().get();
  }

  auto setup = std::make_shared<KeypointAdjustmentSetup>();
  for (int c = 0; c < n_const; ++c) {
    const auto it = graph.image_name_to_id.find("im" + std::to_string(const_images[c]));
    if (it != graph.image_name_to_id.end()) setup->SetImageConstant(it->second);   // an image without matches is not in the graph
  }
  if (const_roots) setup->SetMaskedNodesConstant(&graph, roots);

  FeatureMetricKeypointOptimizer::Options options;
  options.weight_by_sim = weight_by_sim != 0;
  options.root_edges_only = root_edges_only != 0;
  options.root_regularize_weight = root_regularize_weight;
  options.bound = bound;
  options.print_summary = false;
  options.solver_options.minimizer_progress_to_stdout = false;
  InterpolationConfig icfg;
  Recorder opt(options, setup, icfg);

  std::unordered_set<size_t> in_problem;
  if (nodes_in_problem) for (int64_t i = 0; i < n_in; ++i) in_problem.insert((size_t)nodes_in_problem[i]);
  else for (size_t i = 0; i < n; ++i) in_problem.insert(i);
  opt.TopologicalKeypointOptimizer<FeatureMetricKeypointOptimizer>::Run<128, 1>(in_problem, &keypoints, &graph, labels, roots, fview);

  // pointer -> node
  std::unordered_map<const double*, size_t> node_of;
  for (size_t i = 0; i < n; ++i) {
    const FeatureNode* node = graph.nodes[i];
    node_of[keypoints.at(graph.image_id_to_name.at(node->image_id)).row(node->feature_idx).data()] = i;
  }
  ceres::Problem* pr = opt.problem();
  if ((int64_t)pr->blocks.size() > max_blocks) return -1;
  for (size_t b = 0; b < pr->blocks.size(); ++b) {
    blk_src[b] = (int64_t)node_of.at(pr->blocks[b].params[0]);
    blk_dst[b] = (int64_t)node_of.at(pr->blocks[b].params[1]);
    blk_w[b] = static_cast<ceres::ScaledLoss*>(pr->blocks[b].loss)->a_;
  }
  const double nan = std::numeric_limits<double>::quiet_NaN();
  for (size_t i = 0; i < n; ++i) { node_const[i] = 0; for (int j = 0; j < 4; ++j) node_bounds[4 * i + j] = nan; }
  for (double* p : pr->constant) node_const[node_of.at(p)] = 1;
  for (const auto& bd : pr->bounds) node_bounds[4 * node_of.at(bd.p) + (bd.upper ? 2 : 0) + bd.index] = bd.value;
  return (int64_t)pr->blocks.size();
} catch (const std::exception& e) {
  std::fprintf(stderr, "pxo_ref_ka_setup: %s\n", e.what());
  return -5;
}

}  // extern "C"

// ref_loc_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN query refinement set-up where it lies under /root/reference and RECORDS the
// problem it hands to Ceres (nothing is solved):
//   pixsfm/localization/src/single_query_keypoint_optimizer.h + query_keypoint_optimizer.h   QKA: RunQuery (the three
//       reference containers), AddFeatureReferenceResidual, ParameterizeKeypoint (box bounds)
//   pixsfm/localization/src/single_query_bundle_optimizer.h + query_bundle_optimizer.h       QBA: RunQuery, ParameterizeQuery
//   pixsfm/localization/src/nearest_references.h                                             FindNearestReferences
//   pixsfm/residuals/src/feature_reference.h, features/src/references.h
// against the recording ceres::Problem whose cost functions can be EVALUATED on plain doubles (ref_stubs/interp/ceres/ceres.h,
// PXO_STUB_EVALUABLE_COST): the query patches hold zeros and l2_normalize is off, so a block's first residual is minus the
// first entry of the reference descriptor it was built with -- every descriptor carries a unique tag there.
// Output: oracle/_ref/libpxo_ref_loc.so.  Nothing of the reference is copied into this repository.
#define PXO_STUB_EVALUABLE_COST 1
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "localization/src/single_query_keypoint_optimizer.h"
#include "localization/src/single_query_bundle_optimizer.h"
#include "localization/src/nearest_references.h"

namespace pixsfm {
template <typename dtype>
FeaturePatch<dtype>::FeaturePatch() : data_ptr_(nullptr) {}
template <typename dtype>
struct ViewPatch : public FeaturePatch<dtype> {
  ViewPatch(const void* data, int H, int W, int C, const int* corner, const double* scale) {
    this->data_ptr_ = const_cast<dtype*>(static_cast<const dtype*>(data));
    this->shape_ = {H, W, C};
    this->corner_[0] = corner[0]; this->corner_[1] = corner[1];
    this->scale_[0] = scale[0]; this->scale_[1] = scale[1];
  }
};
// the optimizers keep their set-up inside RunQuery; these expose the recorded problem of the last call
struct Recorded { std::vector<ceres::Problem::Block> blocks; std::vector<double*> constant; std::vector<ceres::Problem::Bound> bounds;
                  std::vector<double*> quaternion; std::vector<ceres::Problem::Subset> subset; std::vector<std::vector<double>> residuals; };
}  // namespace pixsfm

// the recording Problem is destroyed at the end of RunQuery: SolveProblem is the last thing that sees it, and the stub's
// ceres::Solve receives the problem -- capture there
namespace ceres {
inline pixsfm::Recorded& LastProblem() { static pixsfm::Recorded r; return r; }
}  // namespace ceres

namespace {
using namespace pixsfm;
constexpr int C = 128, PS = 16;

struct QueryMap {
  std::vector<half> zeros;
  std::vector<std::unique_ptr<ViewPatch<half>>> owned;
  FeatureMap<half> fmap;
  QueryMap(int n_patches, const int32_t* corners, const double* scales, int sparse) : zeros((size_t)PS * PS * C, half(0.0f)) {
    fmap.sparse = sparse != 0;
    fmap.channels = C;
    for (int q = 0; q < n_patches; ++q) {
      owned.emplace_back(new ViewPatch<half>(zeros.data(), PS, PS, C, corners + 2 * q, scales + 2 * q));
      fmap.patches[(colmap::point2D_t)q] = owned.back().get();
    }
  }
};

DescriptorMatrixXd Tagged(double tag) {
  DescriptorMatrixXd d(1, C);
  d.setZero();
  d(0, 0) = tag;
  d(0, 1) = 0.5;
  return d;
}
// mode 0: one descriptor per correspondence (vector<Ref<DescriptorMatrixXd>>), tag 1000 i + 999
// mode 1: ref_count[i] descriptors per correspondence (vector<vector<DescriptorMatrixXd>>), tags 1000 i + r
// mode 2: Reference objects: descriptor tag 1000 i + 999, ref_count[i] observations with tags 1000 i + r (0 = none kept)
struct Refs3 {
  std::vector<DescriptorMatrixXd> pool;
  std::vector<Eigen::Ref<DescriptorMatrixXd>> single;
  std::vector<std::vector<DescriptorMatrixXd>> lists;
  std::vector<Reference> refs;
  Refs3(int n, int mode, const int32_t* ref_count) {
    if (mode == 0) {
      for (int i = 0; i < n; ++i) pool.push_back(Tagged(1000.0 * i + 999));
      for (int i = 0; i < n; ++i) single.emplace_back(pool[i]);
    } else if (mode == 1) {
      lists.resize(n);
      for (int i = 0; i < n; ++i) for (int r = 0; r < ref_count[i]; ++r) lists[i].push_back(Tagged(1000.0 * i + r));
    } else {
      for (int i = 0; i < n; ++i) {
        ReferenceData data;
        for (int r = 0; r < ref_count[i]; ++r) {
          data.track.AddElement(r, r);
          data.observations.push_back(Tagged(1000.0 * i + r));
          data.costs.push_back(0.0);
        }
        refs.emplace_back(colmap::TrackElement(0, 0), Tagged(1000.0 * i + 999), ref_count[i] ? &data : nullptr);
      }
    }
  }
};

void Capture(ceres::Problem* problem) {
  Recorded& r = ceres::LastProblem();
  r = Recorded();
  r.blocks = problem->blocks; r.constant = problem->constant; r.bounds = problem->bounds;
  r.quaternion = problem->quaternion_manifold; r.subset = problem->subset_manifold;
  for (auto& b : problem->blocks) {
    std::vector<double> res((size_t)b.cost->NumResiduals(), 0.0);
    std::vector<const double*> params(b.params.begin(), b.params.end());
    b.cost->EvaluateValues(params.data(), res.data());
    r.residuals.push_back(res);
  }
}
}  // namespace

// the stub's ceres::Solve is called by SolveProblem with the problem still alive
namespace ceres { void PxoSolveHook(Problem* p) { Capture(p); } }

extern "C" {

// QKA set-up.  keypoints [n][2] (COLMAP image coordinates); query map: n_patches zero patches of 16 x 16 x 128 with corners
// / scales; patch_idxs / inliers may be NULL.  Outputs: blocks in the order they were added (keypoint index, descriptor
// tag); per keypoint the box bounds (NaN = none set).  Returns RunQuery's bool (0 / 1), or < 0 on an exception.
int pxo_ref_qka_setup(int n, const double* keypoints_in, int n_patches, const int32_t* corners, const double* scales, int sparse,
                      double bound, int ref_mode, const int32_t* ref_count, const int32_t* patch_idxs, const uint8_t* inliers,
                      int32_t max_blocks, int32_t* n_blocks, int32_t* blk_kp, double* blk_tag, double* lower, double* upper) try {
  QueryMap qm(n_patches, corners, scales, sparse);
  KeypointMatrixd keypoints(n, 2);
  for (int i = 0; i < 2 * n; ++i) keypoints.data()[i] = keypoints_in[i];
  Refs3 refs(n, ref_mode, ref_count);
  QueryKeypointOptimizerOptions options;
  options.bound = bound;
  options.print_summary = false;
  InterpolationConfig icfg;
  icfg.l2_normalize = false;
  icfg.check_bounds = false;
  SingleQueryKeypointOptimizer opt(options, icfg);
  std::vector<colmap::point2D_t> pidx;
  if (patch_idxs) pidx.assign(patch_idxs, patch_idxs + n);
  std::vector<bool> inl;
  if (inliers) inl.assign(inliers, inliers + n);
  ceres::LastProblem() = Recorded();
  Eigen::Ref<KeypointMatrixd> kref(keypoints);
  bool ok;
  if (ref_mode == 0) ok = opt.RunQuery(kref, qm.fmap, refs.single, patch_idxs ? &pidx : nullptr, inliers ? &inl : nullptr);
  else if (ref_mode == 1) ok = opt.RunQuery(kref, qm.fmap, refs.lists, patch_idxs ? &pidx : nullptr, inliers ? &inl : nullptr);
  else ok = opt.RunQuery(kref, qm.fmap, refs.refs, patch_idxs ? &pidx : nullptr, inliers ? &inl : nullptr);
  const Recorded& r = ceres::LastProblem();
  if ((int)r.blocks.size() > max_blocks) return -2;
  *n_blocks = (int32_t)r.blocks.size();
  for (size_t b = 0; b < r.blocks.size(); ++b) {
    blk_kp[b] = (int32_t)((r.blocks[b].params.at(0) - keypoints.data()) / 2);
    blk_tag[b] = -r.residuals[b].at(0);
  }
  for (int i = 0; i < 2 * n; ++i) lower[i] = upper[i] = std::nan("");
  for (auto& bd : r.bounds) {
    const int i = (int)((bd.p - keypoints.data()) / 2);
    (bd.upper ? upper : lower)[2 * i + bd.index] = bd.value;
  }
  return ok ? 1 : 0;
} catch (...) { return -5; }

// QBA set-up.  Outputs: blocks (correspondence index, descriptor tag); per point 1 if held constant; camera_const: -1 = the
// whole parameter block constant, else the bit mask of the constant parameters (subset manifold; 0 = all free);
// quaternion: 1 if the quaternion manifold was set on qvec.
int pxo_ref_qba_setup(int n, const double* points3D_in, int cam_model, int cam_nparams, const double* cam_params, const double* qvec_in,
                      const double* tvec_in, int n_patches, const int32_t* corners, const double* scales, int refine_focal,
                      int refine_pp, int refine_extra, int ref_mode, const int32_t* ref_count, const int32_t* patch_idxs,
                      const uint8_t* inliers, int32_t max_blocks, int32_t* n_blocks, int32_t* blk_point, double* blk_tag,
                      uint8_t* point_const, int32_t* camera_const, int32_t* quaternion) try {
  QueryMap qm(n_patches, corners, scales, 1);
  std::vector<Eigen::Vector3d> points3D(n);
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) points3D[i][k] = points3D_in[3 * i + k];
  colmap::Camera camera;
  camera.SetModelId(cam_model);
  camera.SetParams(std::vector<double>(cam_params, cam_params + cam_nparams));
  Eigen::Vector4d qvec; Eigen::Vector3d tvec;
  for (int k = 0; k < 4; ++k) qvec[k] = qvec_in[k];
  for (int k = 0; k < 3; ++k) tvec[k] = tvec_in[k];
  Refs3 refs(n, ref_mode, ref_count);
  QueryBundleOptimizerOptions options;
  options.refine_focal_length = refine_focal != 0;
  options.refine_principal_point = refine_pp != 0;
  options.refine_extra_params = refine_extra != 0;
  options.print_summary = false;
  InterpolationConfig icfg;
  icfg.l2_normalize = false;
  icfg.check_bounds = false;
  SingleQueryBundleOptimizer opt(options, icfg);
  std::vector<colmap::point2D_t> pidx;
  if (patch_idxs) pidx.assign(patch_idxs, patch_idxs + n);
  std::vector<bool> inl;
  if (inliers) inl.assign(inliers, inliers + n);
  ceres::LastProblem() = Recorded();
  Eigen::Ref<Eigen::Vector4d> qref(qvec);
  Eigen::Ref<Eigen::Vector3d> tref(tvec);
  bool ok;
  if (ref_mode == 0) ok = opt.RunQuery(qref, tref, camera, points3D, qm.fmap, refs.single, patch_idxs ? &pidx : nullptr, inliers ? &inl : nullptr);
  else if (ref_mode == 1) ok = opt.RunQuery(qref, tref, camera, points3D, qm.fmap, refs.lists, patch_idxs ? &pidx : nullptr, inliers ? &inl : nullptr);
  else ok = opt.RunQuery(qref, tref, camera, points3D, qm.fmap, refs.refs, patch_idxs ? &pidx : nullptr, inliers ? &inl : nullptr);
  const Recorded& r = ceres::LastProblem();
  if ((int)r.blocks.size() > max_blocks) return -2;
  *n_blocks = (int32_t)r.blocks.size();
  for (size_t b = 0; b < r.blocks.size(); ++b) {
    const auto& pr = r.blocks[b].params;                      // qvec, tvec, xyz, camera
    if (pr.size() != 4 || pr[0] != qvec.data() || pr[1] != tvec.data() || pr[3] != camera.ParamsData()) return -3;
    int idx = -1;
    for (int i = 0; i < n; ++i) if (pr[2] == points3D[i].data()) idx = i;
    blk_point[b] = idx;
    blk_tag[b] = -r.residuals[b].at(0);
  }
  for (int i = 0; i < n; ++i) point_const[i] = 0;
  *camera_const = 0;
  for (double* p : r.constant) {
    if (p == camera.ParamsData()) { *camera_const = -1; continue; }
    for (int i = 0; i < n; ++i) if (p == points3D[i].data()) point_const[i] = 1;
  }
  for (auto& sb : r.subset) if (sb.p == camera.ParamsData()) for (int c : sb.constant) *camera_const |= 1 << c;
  *quaternion = 0;
  for (double* p : r.quaternion) if (p == qvec.data()) *quaternion = 1;
  return ok ? 1 : 0;
} catch (...) { return -5; }

// FindNearestReferences (nearest_references.h:20-52): query descriptors at `keypoints` in patches [n][16][16][128] (half),
// candidates: cand_count[i] observation descriptors per keypoint, rows of cand_desc; out: index of the chosen one per keypoint
// (within its own candidates) and the descriptor.
int pxo_ref_nearest_references(int n, const double* keypoints_in, const void* patches, const int32_t* corners, const double* scales,
                               const int32_t* cand_count, const double* cand_desc, int l2_normalize, int32_t* chosen, double* out_desc) try {
  FeatureMap<half> fmap;
  fmap.sparse = true; fmap.channels = C;
  std::vector<std::unique_ptr<ViewPatch<half>>> owned;
  for (int q = 0; q < n; ++q) {
    owned.emplace_back(new ViewPatch<half>(static_cast<const half*>(patches) + (size_t)q * PS * PS * C, PS, PS, C, corners + 2 * q, scales + 2 * q));
    fmap.patches[(colmap::point2D_t)q] = owned.back().get();
  }
  std::unordered_map<colmap::point3D_t, Reference> references;
  std::vector<colmap::point3D_t> ids;
  size_t row = 0;
  for (int i = 0; i < n; ++i) {
    ReferenceData data;
    for (int r = 0; r < cand_count[i]; ++r, ++row) {
      DescriptorMatrixXd d(1, C);
      for (int c = 0; c < C; ++c) d(0, c) = cand_desc[row * C + c];
      data.track.AddElement(r, r);
      data.observations.push_back(d);
      data.costs.push_back(0.0);
    }
    references.emplace((colmap::point3D_t)(100 + i), Reference(colmap::TrackElement(0, 0), Tagged(0.0), &data));
    ids.push_back((colmap::point3D_t)(100 + i));
  }
  Eigen::Matrix<double, -1, 2, Eigen::RowMajor> keypoints(n, 2);
  for (int i = 0; i < 2 * n; ++i) keypoints.data()[i] = keypoints_in[i];
  Eigen::Ref<Eigen::Matrix<double, -1, 2, Eigen::RowMajor>> kref(keypoints);
  InterpolationConfig icfg;
  icfg.l2_normalize = l2_normalize != 0;
  std::vector<DescriptorMatrixXd> nearest = FindNearestReferences(fmap, references, kref, ids, icfg, (std::vector<colmap::point3D_t>*)nullptr);
  row = 0;
  for (int i = 0; i < n; ++i) {
    chosen[i] = -1;
    for (int c = 0; c < C; ++c) out_desc[(size_t)i * C + c] = nearest[i](0, c);
    for (int r = 0; r < cand_count[i]; ++r, ++row)
      if (chosen[i] < 0 && std::memcmp(&cand_desc[row * C], &out_desc[(size_t)i * C], sizeof(double) * C) == 0) chosen[i] = r;
  }
  return 0;
} catch (...) { return -5; }
}

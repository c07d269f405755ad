// ref_ba_setup_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN bundle-adjustment problem construction where it lies under /root/reference and RECORDS what it
// hands to Ceres (nothing is solved):
//   pixsfm/bundle_adjustment/src/bundle_optimizer.h                    BundleOptimizer::Run / SetUp / AddImageToProblem /
//       AddPointToProblem / RegisterPoint3DObservation / ParameterizePoints / ParameterizeImages / ParameterizeCameras /
//       SolveProblem's linear-solver selection
//   pixsfm/bundle_adjustment/src/feature_reference_bundle_optimizer.h  AddResiduals (constant-pose vs full functor)
//   pixsfm/bundle_adjustment/src/bundle_adjustment_options.{h,cc}      BundleAdjustmentSetup, BundleOptimizerOptions
// against a recording ceres::Problem, functional stand-ins for the COLMAP scene / BundleAdjustmentConfig classes
// (oracle/ref_stubs/basetup/, restated from the published COLMAP API) and in-memory feature containers.
// Output: oracle/_ref/libpxo_ref_ba_setup.so.  Nothing of the reference is copied into this repository.
#include <cstdint>
#include <memory>
#include <unordered_map>
#include <vector>

#include "bundle_adjustment/src/bundle_adjustment_options.cc"
#include "bundle_adjustment/src/feature_reference_bundle_optimizer.h"

namespace pixsfm {
template <typename dtype>
FeaturePatch<dtype>::FeaturePatch() : data_ptr_(nullptr) {}
struct MetaPatch : public FeaturePatch<half> {
  MetaPatch() { this->shape_ = {16, 16, 128}; this->corner_[0] = 0; this->corner_[1] = 0; this->scale_[0] = 1.0; this->scale_[1] = 1.0; }
};
}  // namespace pixsfm

extern "C" {

// Scene: images 0 .. n_images-1 (camera image_camera[i]; points2D of image i: rows p2d_ptr[i] .. p2d_ptr[i+1] of p2d_point3D, -1
// = no 3D point), cameras (model id; the parameter VALUES are irrelevant to the set-up), points 0 .. n_points-1 whose tracks
// list their observations in (image, point2D) order.
// Setup (BundleAdjustmentSetup): images in the problem, constant-pose images, constant-tvec images with a 3-bit index mask,
// variable / constant extra points, constant cameras.  Options: refine_* flags, min_track_length, use_inner_iterations.
// Outputs (capacity max_blocks): per residual block image, point2D index and whether the constant-pose functor was used (2
// parameter blocks); per image: 0 no pose block in the problem / untouched, 1 constant, 2 quaternion manifold (+ tvec_mask);
// per camera: -1 not in the problem, else bit mask of constant parameters (all bits = block constant); per point: -1 not in
// the problem, 0 variable, 1 constant; inner_group: 1 if in inner-iteration group 0; solver: [linear_solver_type, preconditioner].
int64_t pxo_ref_ba_setup(int n_images, const int32_t* image_camera, const int64_t* p2d_ptr, const int64_t* p2d_point3D,
                         int n_cameras, const int32_t* cam_model, int64_t n_points, const uint8_t* in_problem,
                         const uint8_t* const_pose, const uint8_t* tvec_mask_in, const uint8_t* var_point, const uint8_t* const_point,
                         const uint8_t* const_camera, int refine_focal, int refine_pp, int refine_extra, int refine_extrinsics,
                         int min_track_length, int use_inner, int64_t max_blocks, int32_t* blk_image, int32_t* blk_p2d,
                         uint8_t* blk_const_pose, int8_t* image_role, uint8_t* tvec_mask_out, int32_t* camera_mask,
                         int8_t* point_role, uint8_t* inner_group, int32_t* solver) try {
  using namespace pixsfm;
  static const int kNumParams[5] = {3, 4, 4, 5, 8};
  colmap::Reconstruction rec;
  for (int c = 0; c < n_cameras; ++c) {
    colmap::Camera cam;
    cam.SetModelId(cam_model[c]);
    cam.SetParams(std::vector<double>(kNumParams[cam_model[c]], 1.0));
    rec.cameras_[c] = cam;
  }
  for (int64_t p = 0; p < n_points; ++p) rec.points3D_[p] = colmap::Point3D();
  FeatureView<half> fview;
  MetaPatch patch;
  for (int i = 0; i < n_images; ++i) {
    colmap::Image im;
    im.SetCameraId(image_camera[i]);
    im.Qvec()[0] = 2.0; im.Qvec()[1] = 0.0; im.Qvec()[2] = 0.0; im.Qvec()[3] = 0.0;
    im.Points2D().resize(p2d_ptr[i + 1] - p2d_ptr[i]);
    for (int64_t k = p2d_ptr[i]; k < p2d_ptr[i + 1]; ++k) {
      const int64_t j = k - p2d_ptr[i];
      if (p2d_point3D[k] >= 0) {
        im.Point2D(j).SetPoint3DId((colmap::point3D_t)p2d_point3D[k]);
        rec.points3D_.at(p2d_point3D[k]).Track().AddElement(i, (colmap::point2D_t)j);
      }
      fview.maps[i].patches[(colmap::point2D_t)j] = &patch;
    }
    rec.images_[i] = im;
  }
  std::unordered_map<colmap::point3D_t, Reference> references;
  for (int64_t p = 0; p < n_points; ++p) { Reference r; r.descriptor = DescriptorMatrixXd(1, 128); references[p] = r; }

  BundleAdjustmentSetup setup;
  for (int i = 0; i < n_images; ++i) if (in_problem[i]) setup.AddImage(i);
  for (int i = 0; i < n_images; ++i) {
    if (const_pose[i]) setup.SetConstantPose(i);
    if (tvec_mask_in[i]) {
      std::vector<int> idxs;
      for (int a = 0; a < 3; ++a) if (tvec_mask_in[i] & (1 << a)) idxs.push_back(a);
      setup.SetConstantTvec(i, idxs);
    }
  }
  for (int64_t p = 0; p < n_points; ++p) {
    if (var_point[p]) setup.AddVariablePoint(p);
    if (const_point[p]) setup.AddConstantPoint(p);
  }
  for (int c = 0; c < n_cameras; ++c) if (const_camera[c]) setup.SetConstantCamera(c);

  BundleOptimizerOptions options;
  options.refine_focal_length = refine_focal != 0;
  options.refine_principal_point = refine_pp != 0;
  options.refine_extra_params = refine_extra != 0;
  options.refine_extrinsics = refine_extrinsics != 0;
  options.min_track_length = min_track_length;
  options.solver_options.use_inner_iterations = use_inner != 0;
  options.solver_options.minimizer_progress_to_stdout = false;
  options.print_summary = false;
  InterpolationConfig icfg;
  FeatureReferenceBundleOptimizer opt(options, setup, icfg);
  ceres::LastSolveOptions() = ceres::Solver::Options();   // SolveProblem returns before the solver when there are no residuals
  try {
    opt.Run(&rec, fview, references);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "pxo_ref_ba_setup: the reference threw: %s\n", e.what());
    return -4;
  }

  ceres::Problem* pr = opt.Problem();
  if ((int64_t)pr->blocks.size() > max_blocks) return -1;
  std::unordered_map<const double*, int> image_of_q, camera_of;
  std::unordered_map<const double*, int64_t> point_of;
  std::unordered_map<const double*, int> image_of_t;
  for (int i = 0; i < n_images; ++i) { image_of_q[rec.images_.at(i).Qvec().data()] = i; image_of_t[rec.images_.at(i).Tvec().data()] = i; }
  for (int c = 0; c < n_cameras; ++c) camera_of[rec.cameras_.at(c).ParamsData()] = c;
  for (int64_t p = 0; p < n_points; ++p) point_of[rec.points3D_.at(p).XYZ().data()] = p;
  for (int i = 0; i < n_images; ++i) { image_role[i] = 0; tvec_mask_out[i] = 0; }
  for (int c = 0; c < n_cameras; ++c) camera_mask[c] = -1;
  for (int64_t p = 0; p < n_points; ++p) { point_role[p] = -1; inner_group[p] = 0; }

  // residual blocks: AddResiduals fetches the observation's patch from the feature view right before it adds the block, so the
  // k-th logged fetch names the observation of the k-th block
  if (fview.calls.size() != pr->blocks.size()) return -2;
  for (size_t b = 0; b < pr->blocks.size(); ++b) {
    const auto& params = pr->blocks[b].params;
    const bool cp = params.size() == 2;
    const int64_t p = point_of.at(cp ? params[0] : params[2]);
    const int cam = camera_of.at(cp ? params[1] : params[3]);
    blk_const_pose[b] = cp ? 1 : 0;
    point_role[p] = 0;
    if (camera_mask[cam] < 0) camera_mask[cam] = 0;
    blk_image[b] = (int32_t)fview.calls[b].first;
    blk_p2d[b] = (int32_t)fview.calls[b].second;
    if (!cp && image_of_q.at(params[0]) != blk_image[b]) return -3;
  }
  for (double* p : pr->constant) {
    if (image_of_q.count(p)) image_role[image_of_q.at(p)] = 1;
    else if (image_of_t.count(p)) {}
    else if (camera_of.count(p)) camera_mask[camera_of.at(p)] = (1 << rec.cameras_.at(camera_of.at(p)).NumParams()) - 1;
    else if (point_role[point_of.at(p)] >= 0) point_role[point_of.at(p)] = 1;
    // (else: SetParameterBlockConstant on a point WITHOUT residual blocks -- a variable point whose observations were all
    // filtered out, bundle_optimizer.h:345-347 / :360-363; real Ceres aborts there, so such scenes are outside the contract)
  }
  for (double* q : pr->quaternion_manifold) image_role[image_of_q.at(q)] = 2;
  for (const auto& s : pr->subset_manifold) {
    int mask = 0;
    for (int a : s.constant) mask |= 1 << a;
    if (image_of_t.count(s.p)) tvec_mask_out[image_of_t.at(s.p)] = (uint8_t)mask;
    else camera_mask[camera_of.at(s.p)] = mask;
  }
  // the optimizer works on a COPY of the options: the inner-iteration ordering it filled is the shared_ptr made in Run
  // (not reachable from here); the stub solver kept the options it was handed
  const ceres::Solver::Options& so = ceres::LastSolveOptions();
  if (so.inner_iteration_ordering)
    for (const auto& e : so.inner_iteration_ordering->elements) if (e.second == 0) inner_group[point_of.at(e.first)] = 1;
  solver[0] = (int)so.linear_solver_type; solver[1] = (int)so.preconditioner_type;
  return (int64_t)pr->blocks.size();
} catch (const std::exception& e) {
  std::fprintf(stderr, "pxo_ref_ba_setup: %s\n", e.what());
  return -5;
}

}  // extern "C"

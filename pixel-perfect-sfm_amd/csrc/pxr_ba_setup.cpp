// pxr_ba_setup.cpp -- native host code: the bundle-adjustment PROBLEM CONSTRUCTION of the reference on a flat scene.
//
// BundleOptimizer::SetUp walks colmap::Reconstruction (bundle_adjustment/src/bundle_optimizer.h:139-165): AddImageToProblem
// for every image of the setup (:247-275), AddPointToProblem for the extra variable / constant points (:279-317: their
// observations in images OUTSIDE the setup), FeatureReferenceBundleOptimizer::AddResiduals per observation
// (feature_reference_bundle_optimizer.h:90-149), then ParameterizePoints / Images / Cameras (:335-453).  A binding dumps
// the reconstruction into the arrays below (what pycolmap exposes per image / per point) and gets back the observation list
// and the constancy masks pxr_ba_solve takes -- the walk itself stays native, like in the reference.
// Checked against the reference's own set-up code compiled in place (tests/golden/ba_setup_ref.npz,
// tests/test_ba_setup_golden.py, tools/fuzz_setup_vs_reference.py).
#include <algorithm>
#include <cstdint>
#include <vector>

#include "pxr_internal.h"

namespace {
// [upstream COLMAP 3.8 camera_models.h] number of parameters and the focal / principal point / extra parameter groups
const int kNumParams[11] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};
int focal_mask(int m) { return (m == 1 || m == 4 || m == 5 || m == 6 || m == 7 || m == 10) ? 0b11 : 0b1; }
int pp_mask(int m) { return (m == 1 || m == 4 || m == 5 || m == 6 || m == 7 || m == 10) ? 0b1100 : 0b110; }
int extra_mask(int m) { return ((1 << kNumParams[m]) - 1) & ~(focal_mask(m) | pp_mask(m)); }
}  // namespace

extern "C" int pxr_ba_build_problem(int32_t n_images, const int32_t* image_camera, const int64_t* p2d_ptr,
                                    const int64_t* p2d_point3D, int32_t n_cameras, const int32_t* cam_model, int64_t n_points,
                                    const int64_t* track_ptr, const int32_t* track_image, const int32_t* track_p2d,
                                    const uint8_t* has_patch, const uint8_t* in_setup, const uint8_t* const_pose,
                                    const uint8_t* tvec_mask, const uint8_t* variable_point, const uint8_t* constant_point,
                                    const uint8_t* constant_camera, int refine_focal_length, int refine_principal_point,
                                    int refine_extra_params, int refine_extrinsics, int min_track_length, int skip_missing_patches,
                                    int64_t* n_obs_out, int32_t* obs_image, int32_t* obs_p2d, int64_t* obs_point,
                                    uint8_t* image_in_problem, uint8_t* pose_is_const, uint8_t* tvec_mask_out,
                                    int32_t* camera_mask, int8_t* point_role) {
  PXR_REQUIRE(n_images >= 0 && n_cameras >= 0 && n_points >= 0 && n_obs_out, "pxr_ba_build_problem: invalid sizes");
  PXR_REQUIRE((n_images == 0 || (image_camera && p2d_ptr && in_setup && const_pose && tvec_mask && image_in_problem && pose_is_const && tvec_mask_out)) &&
                  (n_cameras == 0 || (cam_model && constant_camera && camera_mask)) &&
                  (n_points == 0 || (track_ptr && variable_point && constant_point && point_role)),
              "pxr_ba_build_problem: NULL argument");
  const int64_t n_p2d = n_images ? p2d_ptr[n_images] : 0;
  PXR_REQUIRE(n_p2d == 0 || (p2d_point3D && obs_image && obs_p2d && obs_point), "pxr_ba_build_problem: NULL argument");
  for (int c = 0; c < n_cameras; ++c)
    PXR_REQUIRE(cam_model[c] >= 0 && cam_model[c] <= 10, "pxr_ba_build_problem: unsupported camera model id %d", cam_model[c]);
  auto track_len = [&](int64_t p) { return track_ptr[p + 1] - track_ptr[p]; };

  // residual blocks in the order the reference adds them, one per (image, point2D); registered[p] = track elements of p added
  struct Obs { int32_t image, p2d; int64_t point; };
  std::vector<Obs> obs;
  obs.reserve((size_t)n_p2d);
  std::vector<int64_t> registered((size_t)n_points, 0), cam_residuals((size_t)n_cameras, 0);
  std::vector<uint8_t> cam_const(constant_camera, constant_camera + n_cameras);
  for (int i = 0; i < n_images; ++i) image_in_problem[i] = 0;
  auto add_obs = [&](int32_t im, int32_t k, int64_t p) -> int {
    if (has_patch && !has_patch[p2d_ptr[im] + k]) {
      // the optimiser's feature_view.GetFeaturePatch throws (feature_reference_bundle_optimizer.h:100-108); the extractors
      // skip such observations (GetVisibleObservations, reference_extractor.h:171-213)
      if (skip_missing_patches) return PXR_OK;
      return pxr::set_error(PXR_EINVAL, "pxr_ba_build_problem: no feature patch for observation (image %d, point2D %d) of point3D %lld",
                            im, k, (long long)p);
    }
    obs.push_back({im, k, p});
    ++registered[(size_t)p];
    ++cam_residuals[(size_t)image_camera[im]];
    image_in_problem[im] = 1;
    return PXR_OK;
  };
  for (int im = 0; im < n_images; ++im) {                       // AddImageToProblem, images of the setup in ascending id
    if (!in_setup[im]) continue;
    PXR_REQUIRE(image_camera[im] >= 0 && image_camera[im] < n_cameras, "pxr_ba_build_problem: image %d has camera %d", im, image_camera[im]);
    for (int64_t k = p2d_ptr[im]; k < p2d_ptr[im + 1]; ++k) {
      const int64_t p = p2d_point3D[k];
      if (p < 0) continue;                                      // !HasPoint3D
      PXR_REQUIRE(p < n_points, "pxr_ba_build_problem: point2D refers to point3D %lld", (long long)p);
      if (track_len(p) < min_track_length) continue;            // :266-269
      if (int rc = add_obs(im, (int32_t)(k - p2d_ptr[im]), p)) return rc;
    }
  }
  for (int pass = 0; pass < 2; ++pass) {                        // AddPointToProblem: variable points, then constant points
    const uint8_t* sel = pass == 0 ? variable_point : constant_point;
    for (int64_t p = 0; p < n_points; ++p) {
      if (!sel[p]) continue;
      if (registered[(size_t)p] == track_len(p)) continue;      // fully contained already (:289-291)
      for (int64_t e = track_ptr[p]; e < track_ptr[p + 1]; ++e) {
        const int32_t im = track_image[e];
        PXR_REQUIRE(im >= 0 && im < n_images, "pxr_ba_build_problem: track element refers to image %d", im);
        if (in_setup[im]) continue;                             // added by its image, or filtered there (:294-297)
        // the camera of an image outside the setup is refined only if residuals of setup images use it (:305-309)
        if (cam_residuals[(size_t)image_camera[im]] == 0) cam_const[(size_t)image_camera[im]] = 1;
        if (int rc = add_obs(im, track_p2d[e], p)) return rc;
      }
    }
  }
  // ---- Parameterize* ----------------------------------------------------------------------------------------------------
  for (int64_t p = 0; p < n_points; ++p) {                       // ParameterizePoints :335-364
    if (registered[(size_t)p] == 0) { point_role[p] = -1; continue; }
    const int64_t tl = track_len(p);
    const int64_t need = min_track_length > 0 ? std::min<int64_t>(min_track_length, tl) : tl;
    point_role[p] = (need > registered[(size_t)p] || constant_point[p]) ? 1 : 0;
  }
  for (int im = 0; im < n_images; ++im) {                        // ParameterizeImages :366-397
    tvec_mask_out[im] = 0;
    pose_is_const[im] = (!refine_extrinsics || const_pose[im] || !in_setup[im]) ? 1 : 0;
    if (!pose_is_const[im]) tvec_mask_out[im] = tvec_mask[im] & 7;
  }
  const bool all_const = !refine_focal_length && !refine_principal_point && !refine_extra_params;
  for (int c = 0; c < n_cameras; ++c) {                          // ParameterizeCameras :399-442
    if (cam_residuals[(size_t)c] == 0) { camera_mask[c] = -1; continue; }
    const int m = cam_model[c], full = (1 << kNumParams[m]) - 1;
    if (all_const || cam_const[(size_t)c]) { camera_mask[c] = full; continue; }
    int mask = 0;
    if (!refine_focal_length) mask |= focal_mask(m);
    if (!refine_principal_point) mask |= pp_mask(m);
    if (!refine_extra_params) mask |= extra_mask(m);
    camera_mask[c] = mask;
  }
  // ---- observation order: point-major, inside a point the order of Track().Elements() -- what ComputeReference iterates
  // (reference_extractor.h:239-247: the FIRST minimum in track order) and what keeps a point's reference descriptor in L2
  // across its observations in the residual kernel
  std::vector<int64_t> rank(obs.size());
  {
    // position of (image, p2d) inside its point's track: tracks are short, a linear scan per observation is cheap
    for (size_t o = 0; o < obs.size(); ++o) {
      const int64_t p = obs[o].point;
      int64_t r = (int64_t)1 << 40;
      for (int64_t e = track_ptr[p]; e < track_ptr[p + 1]; ++e)
        if (track_image[e] == obs[o].image && track_p2d[e] == obs[o].p2d) { r = e - track_ptr[p]; break; }
      rank[o] = r;
    }
  }
  std::vector<size_t> order(obs.size());
  for (size_t o = 0; o < obs.size(); ++o) order[o] = o;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    if (obs[a].point != obs[b].point) return obs[a].point < obs[b].point;
    return rank[a] < rank[b];
  });
  for (size_t o = 0; o < obs.size(); ++o) {
    obs_image[o] = obs[order[o]].image; obs_p2d[o] = obs[order[o]].p2d; obs_point[o] = obs[order[o]].point;
  }
  *n_obs_out = (int64_t)obs.size();
  return PXR_OK;
}

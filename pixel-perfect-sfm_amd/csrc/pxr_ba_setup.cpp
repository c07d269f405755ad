// pxr_ba_setup.cpp -- native host code: the bundle-adjustment PROBLEM CONSTRUCTION of the reference on a flat scene.
//
// BundleOptimizer::SetUp walks colmap::Reconstruction (bundle_adjustment/src/bundle_optimizer.h:139-165): AddImageToProblem
// for every image of the setup (:247-275), AddPointToProblem for the extra variable / constant points (:279-317: their
// observations in images OUTSIDE the setup), FeatureReferenceBundleOptimizer::AddResiduals per observation
// (feature_reference_bundle_optimizer.h:90-149), then ParameterizePoints / Images / Cameras (:335-453).  A binding dumps
// the reconstruction into the arrays below (what pycolmap exposes per image / per point) and gets back the observation list
// and the constancy masks pxr_ba_solve takes -- the walk itself stays native, like in the reference.
// Checked against the Python restatement of that set-up code under oracle/ on seeded scenes
// (tests/test_ba_setup.py); PARITY UNPINNED: bundle_optimizer.h needs COLMAP / Ceres and cannot be built here.
#include <algorithm>
#include <cstdint>
#include <thread>
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
  // Linear time: a stable counting sort by point (registered[] is the bucket size), then inside every point's short bucket an
  // insertion sort by the position in the track -- the buckets are independent, so the second step runs on all cores
  // (a comparison sort of the 1M observations of BASELINE configs[2] on one core took 0.12 s, twelve LM iterations).
  const size_t n_o = obs.size();
  std::vector<int64_t> first((size_t)n_points + 1, 0);
  for (int64_t p = 0; p < n_points; ++p) first[(size_t)p + 1] = first[(size_t)p] + registered[(size_t)p];
  {
    std::vector<int64_t> cursor(first.begin(), first.end() - 1);
    for (size_t o = 0; o < n_o; ++o) {                              // insertion order inside a bucket = the order of `obs`
      const int64_t at = cursor[(size_t)obs[o].point]++;
      obs_image[at] = obs[o].image; obs_p2d[at] = obs[o].p2d; obs_point[at] = obs[o].point;
    }
  }
  auto order_points = [&](int64_t p0, int64_t p1) {
    std::vector<int64_t> rk;
    for (int64_t p = p0; p < p1; ++p) {
      const int64_t b = first[(size_t)p], e = first[(size_t)p + 1];
      if (e - b < 2) continue;
      rk.resize((size_t)(e - b));
      for (int64_t o = b; o < e; ++o) {                             // position of (image, p2d) inside the point's track
        int64_t r = (int64_t)1 << 40;
        for (int64_t t = track_ptr[p]; t < track_ptr[p + 1]; ++t)
          if (track_image[t] == obs_image[o] && track_p2d[t] == obs_p2d[o]) { r = t - track_ptr[p]; break; }
        rk[(size_t)(o - b)] = r;
      }
      for (int64_t o = b + 1; o < e; ++o) {                         // stable insertion sort by rank
        const int64_t r = rk[(size_t)(o - b)];
        const int32_t im = obs_image[o], k2 = obs_p2d[o];
        int64_t q = o - 1;
        while (q >= b && rk[(size_t)(q - b)] > r) {
          rk[(size_t)(q + 1 - b)] = rk[(size_t)(q - b)]; obs_image[q + 1] = obs_image[q]; obs_p2d[q + 1] = obs_p2d[q];
          --q;
        }
        rk[(size_t)(q + 1 - b)] = r; obs_image[q + 1] = im; obs_p2d[q + 1] = k2;
      }
    }
  };
  {
    unsigned n_thr = std::thread::hardware_concurrency();
    n_thr = std::max(1u, std::min(n_thr, 32u));
    if (n_o < 65536 || n_thr == 1) {
      order_points(0, n_points);
    } else {
      // ranges of ~equal numbers of observations
      std::vector<std::thread> pool;
      int64_t p_lo = 0;
      for (unsigned t = 0; t < n_thr; ++t) {
        const int64_t want = (int64_t)((n_o * (size_t)(t + 1)) / n_thr);
        const int64_t p_hi = t + 1 == n_thr ? n_points
                                            : (int64_t)(std::upper_bound(first.begin(), first.end(), want) - first.begin()) - 1;
        if (p_hi > p_lo) pool.emplace_back(order_points, p_lo, p_hi);
        p_lo = std::max(p_lo, p_hi);
      }
      for (auto& th : pool) th.join();
    }
  }
  *n_obs_out = (int64_t)obs.size();
  return PXR_OK;
}

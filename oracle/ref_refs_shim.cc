// ref_refs_shim.cc -- ORACLE support (test infrastructure only).
//
// Compiles the reference's OWN reference extraction where it lies under /root/reference and runs it on scenes handed over as
// flat arrays:
//   pixsfm/bundle_adjustment/src/reference_extractor.h   ReferenceExtractor::RunSubset / GetVisibleObservations /
//       ComputeReference / FillDescriptorTrack (dynamic template arguments, the path 128-channel features take)
//   pixsfm/base/src/irls_optim.h                         RobustMeanIRLS
//   pixsfm/features/src/dynamic_patch_interpolator.h, patch_interpolator.h, base/src/interpolation.h, projection.h
//   pixsfm/features/src/references.h                    struct Reference / ReferenceData
// against the small dense-matrix stand-in for Eigen (ref_stubs/interp/Eigen/Core), functional stand-ins for the COLMAP scene
// classes (ref_stubs/basetup/) and in-memory feature containers (ref_stubs/kasetup/).  Loss functions: [upstream Ceres]
// restatements (ref_stubs/interp/ceres/ceres.h).  Output: oracle/_ref/libpxo_ref_refs.so.  Nothing of the reference is copied.
#include <cstdint>
#include <cstring>
#include <memory>
#include <unordered_set>
#include <vector>

#include "bundle_adjustment/src/reference_extractor.h"

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
}  // namespace pixsfm

namespace {
using namespace pixsfm;

template <typename dtype>
int Run(int n_cameras, const int32_t* cam_model, const int32_t* cam_nparams, const double* cam_params, int n_images,
        const int32_t* image_camera, const double* qvec, const double* tvec, int64_t n_points, const double* xyz,
        const int64_t* track_ptr, const int32_t* track_image, const int32_t* track_p2d, const int64_t* track_patch,
        const void* patches, int H, int W, int C, const int32_t* corners, const double* scales, int l2_normalize, int use_float_simd,
        int loss_type, const double* loss_params, int iters, int closest_to_robust_mean, int keep_observations, uint8_t* has_ref,
        int32_t* src_image, int32_t* src_p2d, double* descriptor, int32_t* n_kept, double* obs_desc, double* obs_cost) {
  colmap::Reconstruction rec;
  for (int c = 0; c < n_cameras; ++c) {
    colmap::Camera cam;
    cam.SetModelId(cam_model[c]);
    cam.SetParams(std::vector<double>(cam_params + 8 * c, cam_params + 8 * c + cam_nparams[c]));
    rec.cameras_[c] = cam;
  }
  for (int i = 0; i < n_images; ++i) {
    colmap::Image im;
    im.SetCameraId(image_camera[i]);
    for (int k = 0; k < 4; ++k) im.Qvec()[k] = qvec[4 * i + k];
    for (int k = 0; k < 3; ++k) im.Tvec()[k] = tvec[3 * i + k];
    rec.images_[i] = im;
  }
  FeatureView<dtype> fview;
  fview.channels = C;
  std::vector<std::unique_ptr<ViewPatch<dtype>>> owned;
  const size_t patch_elems = (size_t)H * W * C;
  std::unordered_set<colmap::point3D_t> ids;
  for (int64_t p = 0; p < n_points; ++p) {
    colmap::Point3D pt;
    for (int k = 0; k < 3; ++k) pt.XYZ()[k] = xyz[3 * p + k];
    for (int64_t e = track_ptr[p]; e < track_ptr[p + 1]; ++e) {
      pt.Track().AddElement(track_image[e], track_p2d[e]);
      if (track_patch[e] >= 0) {
        const int64_t q = track_patch[e];
        owned.emplace_back(new ViewPatch<dtype>(static_cast<const dtype*>(patches) + q * patch_elems, H, W, C, corners + 2 * q, scales + 2 * q));
        fview.maps[track_image[e]].patches[(colmap::point2D_t)track_p2d[e]] = owned.back().get();
      }
    }
    rec.points3D_[p] = pt;
    ids.insert((colmap::point3D_t)p);
  }
  std::shared_ptr<ceres::LossFunction> loss;
  switch (loss_type) {
    case 0: loss.reset(new ceres::TrivialLoss()); break;
    case 1: loss.reset(new ceres::CauchyLoss(loss_params[0])); break;
    case 2: loss.reset(new ceres::HuberLoss(loss_params[0])); break;
    default: return -2;
  }
  ReferenceConfig config(loss);
  config.iters = iters;
  config.closest_to_robust_mean = closest_to_robust_mean != 0;
  config.keep_observations = keep_observations != 0;
  InterpolationConfig icfg;
  icfg.l2_normalize = l2_normalize != 0;
  icfg.use_float_simd = use_float_simd != 0;
  ReferenceExtractor extractor(config, icfg);
  Refs refs;
  extractor.RunSubset<-1, -1>(ids, refs, &rec, fview);
  for (int64_t p = 0; p < n_points; ++p) {
    auto it = refs.find((colmap::point3D_t)p);
    has_ref[p] = it != refs.end();
    n_kept[p] = 0;
    if (it == refs.end()) continue;
    const Reference& r = it->second;
    src_image[p] = (int32_t)r.source.image_id;
    src_p2d[p] = (int32_t)r.source.point2D_idx;
    if (r.descriptor.rows() != 1 || r.descriptor.cols() != C) return -3;
    std::memcpy(descriptor + (size_t)p * C, r.descriptor.data(), sizeof(double) * C);
    if (keep_observations) {     // the visible part of the track, in track order
      n_kept[p] = (int32_t)r.observations.size();
      if (r.track.Length() != r.observations.size() || r.costs.size() != r.observations.size()) return -4;
      for (size_t k = 0; k < r.observations.size(); ++k) {
        std::memcpy(obs_desc + ((size_t)track_ptr[p] + k) * C, r.observations[k].data(), sizeof(double) * C);
        obs_cost[track_ptr[p] + k] = r.costs[k];
      }
    }
  }
  return 0;
}
}  // namespace

extern "C" {
// Scene: cameras (COLMAP model id, number of parameters, parameters in rows of 8), images (camera, qvec, tvec), points with
// tracks in CSR form (image, point2D index, and the patch of that keypoint in `patches` or -1 = the view holds no patch
// there: GetVisibleObservations skips it).  patches [n][H][W][C] of dtype 0 half / 1 float / 2 double with corners / scales.
// loss_type 0 trivial, 1 Cauchy(a), 2 Huber(a).  Outputs per point: has_ref (points whose visible track is empty get no
// reference), source image / point2D, the descriptor [C]; with keep_observations the per-observation descriptors and costs
// at the rows of the point's track (first n_kept[p] rows).
int pxo_ref_extract_references(int n_cameras, const int32_t* cam_model, const int32_t* cam_nparams, const double* cam_params,
                               int n_images, const int32_t* image_camera, const double* qvec, const double* tvec, int64_t n_points,
                               const double* xyz, const int64_t* track_ptr, const int32_t* track_image, const int32_t* track_p2d,
                               const int64_t* track_patch, int dtype, const void* patches, int H, int W, int C,
                               const int32_t* corners, const double* scales, int l2_normalize, int use_float_simd, int loss_type,
                               const double* loss_params, int iters, int closest_to_robust_mean, int keep_observations,
                               uint8_t* has_ref, int32_t* src_image, int32_t* src_p2d, double* descriptor, int32_t* n_kept,
                               double* obs_desc, double* obs_cost) try {
#define PXO_ARGS n_cameras, cam_model, cam_nparams, cam_params, n_images, image_camera, qvec, tvec, n_points, xyz, track_ptr, track_image, \
    track_p2d, track_patch, patches, H, W, C, corners, scales, l2_normalize, use_float_simd, loss_type, loss_params, iters,             \
    closest_to_robust_mean, keep_observations, has_ref, src_image, src_p2d, descriptor, n_kept, obs_desc, obs_cost
  if (dtype == 0) return Run<half>(PXO_ARGS);
  if (dtype == 1) return Run<float>(PXO_ARGS);
  if (dtype == 2) return Run<double>(PXO_ARGS);
  return -1;
} catch (...) { return -5; }
}

// Functional stand-in for colmap::BundleAdjustmentConfig and the two manifold helpers ([upstream COLMAP 3.8]
// src/optim/bundle_adjustment.{h,cc}: the container semantics restated; the manifold helpers RECORD into the stub Problem).
#pragma once
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "ceres/ceres.h"
#include "colmap/base/reconstruction.h"
namespace colmap {
class BundleAdjustmentConfig {
 public:
  size_t NumImages() const { return image_ids_.size(); }
  void AddImage(image_t i) { image_ids_.insert(i); }
  bool HasImage(image_t i) const { return image_ids_.count(i) > 0; }
  void RemoveImage(image_t i) { image_ids_.erase(i); }
  void SetConstantCamera(camera_t c) { constant_camera_ids_.insert(c); }
  void SetVariableCamera(camera_t c) { constant_camera_ids_.erase(c); }
  bool IsConstantCamera(camera_t c) const { return constant_camera_ids_.count(c) > 0; }
  void SetConstantPose(image_t i) { constant_poses_.insert(i); }
  void SetVariablePose(image_t i) { constant_poses_.erase(i); }
  bool HasConstantPose(image_t i) const { return constant_poses_.count(i) > 0; }
  void SetConstantTvec(image_t i, const std::vector<int>& idxs) { constant_tvecs_.emplace(i, idxs); }
  void RemoveConstantTvec(image_t i) { constant_tvecs_.erase(i); }
  bool HasConstantTvec(image_t i) const { return constant_tvecs_.count(i) > 0; }
  const std::vector<int>& ConstantTvec(image_t i) const { return constant_tvecs_.at(i); }
  void AddVariablePoint(point3D_t p) { variable_point3D_ids_.insert(p); }
  void AddConstantPoint(point3D_t p) { constant_point3D_ids_.insert(p); }
  bool HasPoint(point3D_t p) const { return HasVariablePoint(p) || HasConstantPoint(p); }
  bool HasVariablePoint(point3D_t p) const { return variable_point3D_ids_.count(p) > 0; }
  bool HasConstantPoint(point3D_t p) const { return constant_point3D_ids_.count(p) > 0; }
  const std::unordered_set<image_t>& Images() const { return image_ids_; }
  const std::unordered_set<point3D_t>& VariablePoints() const { return variable_point3D_ids_; }
  const std::unordered_set<point3D_t>& ConstantPoints() const { return constant_point3D_ids_; }
 private:
  std::unordered_set<camera_t> constant_camera_ids_;
  std::unordered_set<image_t> image_ids_;
  std::unordered_set<point3D_t> variable_point3D_ids_, constant_point3D_ids_;
  std::unordered_set<image_t> constant_poses_;
  std::unordered_map<image_t, std::vector<int>> constant_tvecs_;
};
inline void SetQuaternionManifold(ceres::Problem* problem, double* qvec) { problem->quaternion_manifold.push_back(qvec); }
inline void SetSubsetManifold(int size, const std::vector<int>& constant_params, ceres::Problem* problem, double* params) {
  problem->subset_manifold.push_back(ceres::Problem::Subset{params, size, constant_params});
}
}  // namespace colmap

// Functional stand-ins for the COLMAP scene classes the reference's bundle-adjustment SET-UP code walks ([upstream COLMAP 3.8]
// API names and semantics: src/base/{reconstruction,image,point2d,point3d,track,camera}.h), built by the shim from flat arrays.
// Test infrastructure: only what bundle_optimizer.h / feature_reference_bundle_optimizer.h call.
#pragma once
#include <cmath>
#include <map>
#include <string>
#include <vector>
#include "Eigen/Core"
#include "colmap/base/camera_models.h"
#include "colmap/util/types.h"
namespace colmap {
const point3D_t kInvalidPoint3DId = (point3D_t)-1;
struct TrackElement { TrackElement() {} TrackElement(image_t i, point2D_t p) : image_id(i), point2D_idx(p) {} image_t image_id = 0; point2D_t point2D_idx = 0; };
class Track {
 public:
  size_t Length() const { return elements_.size(); }
  const std::vector<TrackElement>& Elements() const { return elements_; }
  const TrackElement& Element(size_t i) const { return elements_.at(i); }
  void AddElement(image_t i, point2D_t p) { elements_.emplace_back(i, p); }
 private:
  std::vector<TrackElement> elements_;
};
class Point3D {
 public:
  Eigen::Vector3d& XYZ() { return xyz_; }
  const Eigen::Vector3d& XYZ() const { return xyz_; }
  class Track& Track() { return track_; }
  const class Track& Track() const { return track_; }
 private:
  Eigen::Vector3d xyz_; class Track track_;
};
class Point2D {
 public:
  bool HasPoint3D() const { return point3D_id_ != kInvalidPoint3DId; }
  point3D_t Point3DId() const { return point3D_id_; }
  void SetPoint3DId(point3D_t id) { point3D_id_ = id; }
 private:
  point3D_t point3D_id_ = kInvalidPoint3DId;
};
class Image {
 public:
  camera_t CameraId() const { return camera_id_; }
  void SetCameraId(camera_t c) { camera_id_ = c; }
  void NormalizeQvec() { const double n = qvec_.norm(); for (int i = 0; i < 4; ++i) qvec_[i] /= n; }   // [upstream] NormalizeQuaternion
  point2D_t NumPoints2D() const { return (point2D_t)points2D_.size(); }
  class Point2D& Point2D(point2D_t i) { return points2D_.at(i); }
  const class Point2D& Point2D(point2D_t i) const { return points2D_.at(i); }
  std::vector<class Point2D>& Points2D() { return points2D_; }
  Eigen::Vector4d& Qvec() { return qvec_; }
  Eigen::Vector3d& Tvec() { return tvec_; }
  const Eigen::Vector4d& Qvec() const { return qvec_; }
  const Eigen::Vector3d& Tvec() const { return tvec_; }
  Eigen::Matrix3d RotationMatrix() const { Eigen::stub_unreachable("Image::RotationMatrix"); }
  const std::string& Name() const { return name_; }
  Eigen::Matrix<double, 3, 4> ProjectionMatrix() const { return Eigen::Matrix<double, 3, 4>(); }
 private:
  camera_t camera_id_ = 0; Eigen::Vector4d qvec_; Eigen::Vector3d tvec_; std::vector<class Point2D> points2D_; std::string name_;
};
class Reconstruction {
 public:
  class Image& Image(image_t i) { return images_.at(i); }
  const class Image& Image(image_t i) const { return images_.at(i); }
  class Camera& Camera(camera_t i) { return cameras_.at(i); }
  const class Camera& Camera(camera_t i) const { return cameras_.at(i); }
  class Point3D& Point3D(point3D_t i) { return points3D_.at(i); }
  const class Point3D& Point3D(point3D_t i) const { return points3D_.at(i); }
  std::map<image_t, class Image> images_; std::map<camera_t, class Camera> cameras_; std::map<point3D_t, class Point3D> points3D_;
};
class Timer { public: void Start() {} void Pause() {} double ElapsedSeconds() const { return 0.0; } };
inline int GetEffectiveNumThreads(int n) { return n > 0 ? n : 1; }
inline Eigen::Vector2d ProjectPointToImage(const Eigen::Vector3d&, const Eigen::Matrix<double, 3, 4>&, const Camera&) { return Eigen::Vector2d(); }
}  // namespace colmap

// Stub for the COLMAP types bundle_adjustment/src/costmap_extractor.h names in its (un-instantiated) driver templates:
// declarations only.  FillPointCostmap, the function built and run by oracle/ref_costmap_shim.cc, touches none of them.
#pragma once
#include <string>
#include <vector>
#include "Eigen/Core"
#include "colmap/base/camera_models.h"
#include "colmap/util/types.h"
namespace colmap {
struct TrackElement { TrackElement() {} TrackElement(image_t i, point2D_t p) : image_id(i), point2D_idx(p) {} image_t image_id = 0; point2D_t point2D_idx = 0; };
class Track { public: const std::vector<TrackElement>& Elements() const; };
class Point3D { public: const class Track& Track() const; const Eigen::Vector3d& XYZ() const; };
class Point2D { public: bool HasPoint3D() const; point3D_t Point3DId() const; };
class Image {
 public:
  const class Point2D& Point2D(point2D_t) const; camera_t CameraId() const; const std::string& Name() const;
  Eigen::Matrix<double, 3, 4> ProjectionMatrix() const;
};
class Reconstruction {
 public:
  const class Point3D& Point3D(point3D_t) const; const class Image& Image(image_t) const; const class Camera& Camera(camera_t) const;
};
class Timer { public: void Start() {} void Pause() {} double ElapsedSeconds() const { return 0.0; } };
inline int GetEffectiveNumThreads(int n) { return n > 0 ? n : 1; }
Eigen::Vector2d ProjectPointToImage(const Eigen::Vector3d&, const Eigen::Matrix<double, 3, 4>&, const Camera&);
}  // namespace colmap

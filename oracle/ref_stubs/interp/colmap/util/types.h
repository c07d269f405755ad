#pragma once
#include <cstdint>
namespace colmap { typedef uint32_t image_t; typedef uint32_t point2D_t; typedef uint64_t point3D_t; typedef uint32_t camera_t; }

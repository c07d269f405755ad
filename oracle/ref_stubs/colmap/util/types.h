// Stub for <colmap/util/types.h> (COLMAP 3.8 is absent in this image): base/src/graph.{h,cc} only
// use the two id typedefs [upstream COLMAP: both uint32_t] and expect the standard containers
// the real header pulls in.
#pragma once
#include <cstdint>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>
namespace colmap {
typedef uint32_t image_t;
typedef uint32_t point2D_t;
}  // namespace colmap

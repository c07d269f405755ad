// Shadows the reference's util/src/types.h (which pulls Eigen, COLMAP and pybind11 in): the aliases the headers built in
// place use.
#pragma once
#include <string>
#include <unordered_map>
#include "Eigen/Core"
#include "colmap/util/types.h"
#include "third-party/half.hpp"
#include "pybind11/pybind11.h"
#include "util/src/log_exceptions.h"
using half = half_float::half;
namespace pixsfm {
template <typename T> using VectorX = Eigen::Matrix<T, Eigen::Dynamic, 1>;
template <int n_nodes, int channels> using DescriptorMatrixd = Eigen::Matrix<double, n_nodes, channels, Eigen::RowMajor>;
using DescriptorMatrixXd = DescriptorMatrixd<Eigen::Dynamic, Eigen::Dynamic>;
template <int n_nodes> using OffsetMatrix3d = Eigen::Matrix<double, n_nodes, 3, Eigen::RowMajor>;
template <int rows> using VectorNd = Eigen::Matrix<double, rows, 1>;
const colmap::point2D_t kDensePatchId = 1000000;
using KeypointMatrixd = Eigen::Matrix<double, -1, 2, Eigen::RowMajor>;
using MapNameKeypoints = std::unordered_map<std::string, KeypointMatrixd>;
}  // namespace pixsfm

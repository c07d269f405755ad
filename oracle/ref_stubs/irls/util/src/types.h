// Stub shadowing the reference's util/src/types.h for the in-place build of base/src/irls_optim.h: Eigen is absent in
// this image, so DescriptorMatrixd and the handful of Eigen expressions RobustMeanIRLS is written with
// (irls_optim.h:29-68) are provided by a minimal row-major matrix class with plain left-to-right loops.  What this
// pins is the reference's CONTROL FLOW (weight normalisation, row normalisation, weights from the loss VALUE, the early
// return); real Eigen reduces sums in packets, so results agree with a real build only to rounding (~1e-15).
#pragma once
#include <cmath>
#include <vector>
#include "util/src/log_exceptions.h"
namespace Eigen {
constexpr int Dynamic = -1;
constexpr int RowMajor = 1;
struct MiniExpr;
class MiniMat {
 public:
  MiniMat() {}
  MiniMat(int r, int c) : r_(r), c_(c), v_((size_t)r * c, 0.0) {}
  int rows() const { return r_; }
  int cols() const { return c_; }
  int size() const { return r_ * c_; }
  void resize(int r, int c) { r_ = r; c_ = c; v_.assign((size_t)r * c, 0.0); }
  void setZero() { for (double& x : v_) x = 0.0; }
  double* data() { return v_.data(); }
  const double* data() const { return v_.data(); }
  double sum() const { double s = 0.0; for (double x : v_) s += x; return s; }
  double squaredNorm() const { double s = 0.0; for (double x : v_) s += x * x; return s; }
  double& operator()(int i) { return v_[i]; }
  double& operator[](int i) { return v_[i]; }
  double operator[](int i) const { return v_[i]; }
  MiniMat operator*(double w) const { MiniMat o(*this); for (double& x : o.v_) x *= w; return o; }
  MiniMat operator/(double w) const { MiniMat o(*this); for (double& x : o.v_) x /= w; return o; }
  MiniMat operator-(const MiniMat& b) const { MiniMat o(*this); for (size_t i = 0; i < v_.size(); ++i) o.v_[i] -= b.v_[i]; return o; }
  MiniMat& operator+=(const MiniMat& b) { for (size_t i = 0; i < v_.size(); ++i) v_[i] += b.v_[i]; return *this; }
  MiniMat transpose() const { MiniMat o(*this); o.r_ = c_; o.c_ = r_; return o; }   // vectors only
  struct RowRef {
    MiniMat* m; int j;
    void normalize() {
      double s = 0.0;
      for (int c = 0; c < m->c_; ++c) s += m->v_[(size_t)j * m->c_ + c] * m->v_[(size_t)j * m->c_ + c];
      const double n = std::sqrt(s);
      for (int c = 0; c < m->c_; ++c) m->v_[(size_t)j * m->c_ + c] /= n;
    }
  };
  RowRef row(int j) { return RowRef{this, j}; }
  static MiniMat Ones(int n) { MiniMat o(n, 1); for (double& x : o.v_) x = 1.0; return o; }
 private:
  int r_ = 0, c_ = 0;
  std::vector<double> v_;
};
template <typename T, int R, int C, int Opt = 0>
using Matrix = MiniMat;
using VectorXd = MiniMat;
}  // namespace Eigen
namespace pixsfm {
template <int n_nodes, int channels>
using DescriptorMatrixd = Eigen::MiniMat;
}  // namespace pixsfm

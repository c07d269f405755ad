// Stub of <ceres/ceres.h> for compiling base/src/irls_optim.h in place: the ceres::LossFunction interface and the
// loss functions pixsfm can hand to RobustMeanIRLS.  [upstream Ceres 2.1 loss_function.cc: the rho formulas are
// restated here, so what this build pins is the IRLS loop of the reference, not Ceres' robustifiers.]
#pragma once
#include <cmath>
#include <limits>
namespace ceres {
class LossFunction {
 public:
  virtual ~LossFunction() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class TrivialLoss : public LossFunction {
 public:
  void Evaluate(double s, double rho[3]) const override { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
};
class CauchyLoss : public LossFunction {
 public:
  explicit CauchyLoss(double a) : b_(a * a), c_(1 / b_) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_;
    const double inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::fmax(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }
 private:
  const double b_, c_;
};
class HuberLoss : public LossFunction {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::fmax(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
  }
 private:
  const double a_, b_;
};
}  // namespace ceres

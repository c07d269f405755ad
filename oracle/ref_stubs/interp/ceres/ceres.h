// Stub for <ceres/ceres.h>: glog's CHECK_* macros, ceres::sqrt, the Jet type's NAME (the shim differentiates with the
// reference's analytic derivative path, Jets are never instantiated), and the two Ceres interpolation entry points the
// reference falls back to below 8 channels / in CERES_BICUBIC mode (declared, aborting: not on the tested path).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <iostream>
#include <memory>
#include <utility>
#include <string>
#include <vector>
#include "Eigen/Core"
#define CERES_VERSION_MAJOR 2
#define PXO_STUB_CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s\n", #cond); std::abort(); } } while (0)
#define CHECK_GE(a, b) PXO_STUB_CHECK((a) >= (b))
#define CHECK_LT(a, b) PXO_STUB_CHECK((a) < (b))
#define CHECK_EQ(a, b) PXO_STUB_CHECK((a) == (b))
#define CHECK_NOTNULL(a) (a)
namespace pxo_stub { struct CheckStream { explicit CheckStream(bool ok) { if (!ok) { std::fprintf(stderr, "CHECK failed\n"); std::abort(); } } template <typename T> CheckStream& operator<<(const T&) { return *this; } }; }
#define CHECK(c) pxo_stub::CheckStream(static_cast<bool>(c))
namespace ceres {
using std::sqrt;
// A forward-mode dual number standing in for ceres::Jet ([upstream Ceres 2.x] jet.h semantics: value a, derivative vector v;
// product / quotient / sqrt rules), enough for the reference's coordinate transform, bounds check, projection and Jet bridge.
template <typename T, int N> struct Jet {
  T a; Eigen::Matrix<T, N, 1> v;
  Jet() : a(T(0)) { for (int i = 0; i < N; ++i) v[i] = T(0); }
  Jet(const T& x) : a(x) { for (int i = 0; i < N; ++i) v[i] = T(0); }     // NOLINT: T(1.0) literals in the reference's templates
  Jet(const T& x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = T(i == k ? 1 : 0); }
  Jet& operator+=(const Jet& y) { a += y.a; v += y.v; return *this; }
  Jet& operator-=(const Jet& y) { a -= y.a; v -= y.v; return *this; }
  Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
  Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
};
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& x) { Jet<T, N> o; o.a = -x.a; o.v = x.v * T(-1); return o; }
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N>& x, const Jet<T, N>& y) { Jet<T, N> o; o.a = x.a + y.a; o.v = x.v + y.v; return o; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& x, const Jet<T, N>& y) { Jet<T, N> o; o.a = x.a - y.a; o.v = x.v - y.v; return o; }
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N>& x, const Jet<T, N>& y) { Jet<T, N> o; o.a = x.a * y.a; o.v = y.a * x.v + x.a * y.v; return o; }
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N>& x, const Jet<T, N>& y) {
  Jet<T, N> o; const T inv = T(1) / y.a, q = x.a * inv; o.a = q; o.v = (x.v - q * y.v) * inv; return o;
}
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N>& x, const T& s) { Jet<T, N> o(x); o.a = x.a + s; return o; }
template <typename T, int N> Jet<T, N> operator+(const T& s, const Jet<T, N>& x) { return x + s; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& x, const T& s) { Jet<T, N> o(x); o.a = x.a - s; return o; }
template <typename T, int N> Jet<T, N> operator-(const T& s, const Jet<T, N>& x) { Jet<T, N> o = -x; o.a = s - x.a; return o; }
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N>& x, const T& s) { Jet<T, N> o; o.a = x.a * s; o.v = x.v * s; return o; }
template <typename T, int N> Jet<T, N> operator*(const T& s, const Jet<T, N>& x) { return x * s; }
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N>& x, const T& s) { Jet<T, N> o; const T inv = T(1) / s; o.a = x.a * inv; o.v = x.v * inv; return o; }
template <typename T, int N> Jet<T, N> operator/(const T& s, const Jet<T, N>& x) { return Jet<T, N>(s) / x; }
template <typename T, int N> bool operator>(const Jet<T, N>& x, const T& s) { return x.a > s; }
template <typename T, int N> bool operator<(const Jet<T, N>& x, const T& s) { return x.a < s; }
template <typename T, int N> bool operator<(const Jet<T, N>& x, const Jet<T, N>& y) { return x.a < y.a; }
template <typename T, int N> bool operator>(const Jet<T, N>& x, const Jet<T, N>& y) { return x.a > y.a; }
template <typename T, int N> Jet<T, N> sqrt(const Jet<T, N>& x) { Jet<T, N> o; o.a = std::sqrt(x.a); o.v = x.v * (T(1) / (T(2) * o.a)); return o; }
template <typename T, int N> Jet<T, N> abs(const Jet<T, N>& x) { return x.a < T(0) ? -x : x; }
using std::abs;
// the factory plumbing the residual headers name in their static Create() members (never called by the shims)
class CostFunction {
 public:
  virtual ~CostFunction() {}
#ifdef PXO_STUB_EVALUABLE_COST   // shims that identify a recorded residual block by evaluating it on plain doubles
  virtual int NumResiduals() const { return 0; }
  virtual bool EvaluateValues(double const* const*, double*) const { return false; }
#endif
};
// [upstream Ceres 2.1 loss_function.cc]: the rho formulas restated (what a build of the reference's cost-map extraction pins is
// its own arithmetic around them, not Ceres' robustifiers)
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
template <typename Functor, int kNumResiduals, int... Ns>
class AutoDiffCostFunction : public CostFunction {
 public:
  explicit AutoDiffCostFunction(Functor* f) : f_(f) {}
#ifdef PXO_STUB_EVALUABLE_COST
  int NumResiduals() const override { return kNumResiduals; }
  bool EvaluateValues(double const* const* p, double* r) const override { return Call(p, r, std::make_index_sequence<sizeof...(Ns)>()); }
 private:
  template <size_t... I> bool Call(double const* const* p, double* r, std::index_sequence<I...>) const { return (*f_)(p[I]..., r); }
#endif
 private:
  std::unique_ptr<Functor> f_;
};
template <int kDataDimension, typename... A>
void CubicHermiteSpline(const A&...) { Eigen::stub_unreachable("ceres::CubicHermiteSpline"); }
template <typename Grid>
class BiCubicInterpolator {
 public:
  explicit BiCubicInterpolator(const Grid&) {}
  void Evaluate(double, double, double*, double*, double*) const { Eigen::stub_unreachable("ceres::BiCubicInterpolator"); }
};

// ---- a RECORDING ceres::Problem / Solver: what the reference's problem set-up code hands to Ceres is kept for inspection,
// nothing is solved ([upstream Ceres] API names only)
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum LoggingType { SILENT, PER_MINIMIZER_ITERATION };
enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL };
class ParameterBlockOrdering { public: void AddElementToGroup(double* p, int g) { elements.push_back(std::make_pair(p, g)); } std::vector<std::pair<double*, int>> elements; };
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
class ScaledLoss : public LossFunction {
 public:
  ScaledLoss(const LossFunction* rho, double a, Ownership) : rho_(rho), a_(a) {}
  void Evaluate(double s, double out[3]) const override {
    if (!rho_) { out[0] = a_ * s; out[1] = a_; out[2] = 0.0; return; }
    rho_->Evaluate(s, out); out[0] *= a_; out[1] *= a_; out[2] *= a_;
  }
  const LossFunction* rho_; double a_;
};
struct IterationSummary {
  int iteration = 0; double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, trust_region_radius = 0;
  int linear_solver_iterations = 0; double iteration_time_in_seconds = 0, cumulative_time_in_seconds = 0;
};
class IterationCallback { public: virtual ~IterationCallback() {} virtual CallbackReturnType operator()(const IterationSummary&) { return SOLVER_CONTINUE; } };
typedef void* ResidualBlockId;
class Problem {
 public:
  struct Options { Ownership loss_function_ownership = TAKE_OWNERSHIP, cost_function_ownership = TAKE_OWNERSHIP; };
  struct Block { CostFunction* cost; LossFunction* loss; std::vector<double*> params; };
  struct Bound { double* p; int index; double value; bool upper; };
  struct Subset { double* p; int size; std::vector<int> constant; };
  std::vector<double*> quaternion_manifold; std::vector<Subset> subset_manifold;
  int NumResiduals() const { return (int)blocks.size(); }
  Problem() {}
  explicit Problem(const Options&) {}
  template <typename... Ps>
  ResidualBlockId AddResidualBlock(CostFunction* c, LossFunction* l, Ps*... ps) { blocks.push_back(Block{c, l, {ps...}}); return nullptr; }
  void SetParameterBlockConstant(double* p) { constant.push_back(p); }
  void SetParameterLowerBound(double* p, int i, double v) { bounds.push_back(Bound{p, i, v, false}); }
  void SetParameterUpperBound(double* p, int i, double v) { bounds.push_back(Bound{p, i, v, true}); }
  std::vector<Block> blocks; std::vector<double*> constant; std::vector<Bound> bounds;
};
class Solver {
 public:
  struct Options {
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY; int max_num_iterations = 50; bool minimizer_progress_to_stdout = false;
    int max_num_consecutive_invalid_steps = 5; double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    int num_threads = 1; LoggingType logging_type = PER_MINIMIZER_ITERATION; std::vector<IterationCallback*> callbacks;
    bool use_inner_iterations = false; std::shared_ptr<ParameterBlockOrdering> inner_iteration_ordering;
    PreconditionerType preconditioner_type = JACOBI; int max_linear_solver_iterations = 500, max_consecutive_nonmonotonic_steps = 5;
    bool IsValid(std::string*) const { return true; }
  };
  struct Summary {
    double initial_cost = 0, final_cost = 0, total_time_in_seconds = 0; int num_residuals_reduced = 1; std::vector<IterationSummary> iterations;
  };
};
// the options the reference finally hands to the solver (linear solver / preconditioner selection) are kept for inspection
inline Solver::Options& LastSolveOptions() { static Solver::Options o; return o; }
#ifdef PXO_STUB_EVALUABLE_COST
void PxoSolveHook(Problem* p);   // defined by the shim: looks at the problem while it is still alive
inline void Solve(const Solver::Options& o, Problem* p, Solver::Summary*) { LastSolveOptions() = o; PxoSolveHook(p); }
#else
inline void Solve(const Solver::Options& o, Problem*, Solver::Summary*) { LastSolveOptions() = o; }
#endif
}  // namespace ceres

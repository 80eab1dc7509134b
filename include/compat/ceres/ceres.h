// ceres-named compatibility surface: exactly the ceres:: names the reference's call sites use
// (SURVEY.md §8b), so that app/main.cpp and include/bundleadjust/*.h read like the reference.
// Ceres itself is NOT here (un-vendored submodule of the reference, /root/reference/.gitmodules:4-6);
// nothing in this header solves anything: ceres::Solve is not provided, BundleAdjustment::solve()
// calls libmiba (include/miba.h) instead.
//
//   Solver::Options fields        app/main.cpp:21-28, src/BundleAdjustment.cpp:104-105
//   LinearSolverTypeToString      src/BundleAdjustment.cpp:20,109
//   IterationCallback/Summary     src/MeshWriterCallback.h:11-23, src/BundleAdjustment.cpp:100-102
//   Solver::Summary::FullReport   src/BundleAdjustment.cpp:110
//   CostFunction / AutoDiffCostFunction / Jet    src/BAConstraint.cpp:82-86
#pragma once

#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

namespace ceres {

enum LinearSolverType {
  DENSE_NORMAL_CHOLESKY = 0, DENSE_QR = 1, SPARSE_NORMAL_CHOLESKY = 2, DENSE_SCHUR = 3,
  SPARSE_SCHUR = 4, ITERATIVE_SCHUR = 5, CGNR = 6
};
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT = 0, DOGLEG = 1 };
enum PreconditionerType { IDENTITY = 0, JACOBI = 1, SCHUR_JACOBI = 2 };
enum CallbackReturnType { SOLVER_CONTINUE = 0, SOLVER_ABORT = 1, SOLVER_TERMINATE_SUCCESSFULLY = 2 };
enum TerminationType { CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2, USER_SUCCESS = 3, USER_FAILURE = 4 };

inline const char* LinearSolverTypeToString(LinearSolverType t) {
  switch (t) {
    case DENSE_NORMAL_CHOLESKY: return "DENSE_NORMAL_CHOLESKY";
    case DENSE_QR: return "DENSE_QR";
    case SPARSE_NORMAL_CHOLESKY: return "SPARSE_NORMAL_CHOLESKY";
    case DENSE_SCHUR: return "DENSE_SCHUR";
    case SPARSE_SCHUR: return "SPARSE_SCHUR";
    case ITERATIVE_SCHUR: return "ITERATIVE_SCHUR";
    case CGNR: return "CGNR";
  }
  return "UNKNOWN";
}
inline const char* TerminationTypeToString(TerminationType t) {
  switch (t) {
    case CONVERGENCE: return "CONVERGENCE";
    case NO_CONVERGENCE: return "NO_CONVERGENCE";
    case FAILURE: return "FAILURE";
    case USER_SUCCESS: return "USER_SUCCESS";
    case USER_FAILURE: return "USER_FAILURE";
  }
  return "UNKNOWN";
}

struct IterationSummary {
  int iteration = 0;
  bool step_is_valid = false;
  bool step_is_nonmonotonic = false;
  bool step_is_successful = false;
  double cost = 0, cost_change = 0, gradient_max_norm = 0, gradient_norm = 0, step_norm = 0;
  double relative_decrease = 0, trust_region_radius = 0, eta = 0;
  int linear_solver_iterations = 0;
  double iteration_time_in_seconds = 0, cumulative_time_in_seconds = 0;
};

class IterationCallback {
 public:
  virtual ~IterationCallback() {}
  virtual CallbackReturnType operator()(const IterationSummary& summary) = 0;
};

struct Solver {
  struct Options {
    TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
    bool use_nonmonotonic_steps = false;
    LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
    PreconditionerType preconditioner_type = JACOBI;
    bool minimizer_progress_to_stdout = false;
    int max_num_iterations = 50;
    int num_threads = 1;
    double max_solver_time_in_seconds = 1e9;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    int max_num_consecutive_invalid_steps = 5;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    double eta = 1e-1;
    int min_linear_solver_iterations = 0;
    int max_linear_solver_iterations = 500;
    bool jacobi_scaling = true;
    bool update_state_every_iteration = false;
    std::vector<IterationCallback*> callbacks;
  };

  struct Summary {
    TerminationType termination_type = NO_CONVERGENCE;
    std::string message;
    double initial_cost = 0, final_cost = 0;
    int num_successful_steps = 0, num_unsuccessful_steps = 0;
    int num_residual_blocks = 0, num_residuals = 0;
    int num_parameter_blocks = 0, num_parameters = 0;
    int num_linear_solver_iterations = 0;
    double total_time_in_seconds = 0, linear_solver_time_in_seconds = 0;
    double jacobian_evaluation_time_in_seconds = 0, residual_evaluation_time_in_seconds = 0;
    LinearSolverType linear_solver_type_given = SPARSE_NORMAL_CHOLESKY;
    std::string linear_solver_used;
    std::vector<IterationSummary> iterations;

    std::string BriefReport() const {
      char b[256];
      snprintf(b, sizeof(b), "MIBA Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s",
               num_successful_steps + num_unsuccessful_steps, initial_cost, final_cost, TerminationTypeToString(termination_type));
      return b;
    }
    std::string FullReport() const {
      char b[2048];
      snprintf(b, sizeof(b),
               "\nSolver Summary (libmiba, MI355X HIP; v = not Ceres)\n\n"
               "Parameter blocks   %12d\nParameters         %12d\nResidual blocks    %12d\nResiduals          %12d\n\n"
               "Minimizer                        TRUST_REGION\nTrust region strategy     LEVENBERG_MARQUARDT\n"
               "Linear solver (given)   %20s\nLinear solver (used)    %s\n\n"
               "Cost:\nInitial        %16.6e\nFinal          %16.6e\nChange         %16.6e\n\n"
               "Minimizer iterations   %8d\nSuccessful steps       %8d\nUnsuccessful steps     %8d\nLinear solver iterations %6d\n\n"
               "Time (in seconds):\n  Residual evaluation  %10.6f\n  Jacobian evaluation  %10.6f\n  Linear solver        %10.6f\nTotal                  %10.6f\n\n"
               "Termination: %20s (%s)\n",
               num_parameter_blocks, num_parameters, num_residual_blocks, num_residuals,
               LinearSolverTypeToString(linear_solver_type_given), linear_solver_used.c_str(),
               initial_cost, final_cost, initial_cost - final_cost,
               num_successful_steps + num_unsuccessful_steps, num_successful_steps, num_unsuccessful_steps,
               num_linear_solver_iterations, residual_evaluation_time_in_seconds, jacobian_evaluation_time_in_seconds,
               linear_solver_time_in_seconds, total_time_in_seconds, TerminationTypeToString(termination_type), message.c_str());
      return b;
    }
  };
};

// ---- dual numbers, just enough for BAConstraint::operator()<Jet> ---------------------------
template <typename T, int N>
struct Jet {
  T a;
  T v[N];
  Jet() : a(T(0)) { for (int i = 0; i < N; ++i) v[i] = T(0); }
  Jet(const T& s) : a(s) { for (int i = 0; i < N; ++i) v[i] = T(0); }  // NOLINT
  Jet(const T& s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = T(0); v[k] = T(1); }
};
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N>& x, const Jet<T, N>& y) { Jet<T, N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& x, const Jet<T, N>& y) { Jet<T, N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& x) { Jet<T, N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N>& x, const Jet<T, N>& y) { Jet<T, N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N>& x, const Jet<T, N>& y) { Jet<T, N> r; const T inv = T(1) / y.a; r.a = x.a * inv; for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
template <typename T, int N> Jet<T, N> operator+(const Jet<T, N>& x, T s) { Jet<T, N> r = x; r.a += s; return r; }
template <typename T, int N> Jet<T, N> operator+(T s, const Jet<T, N>& x) { return x + s; }
template <typename T, int N> Jet<T, N> operator-(const Jet<T, N>& x, T s) { Jet<T, N> r = x; r.a -= s; return r; }
template <typename T, int N> Jet<T, N> operator-(T s, const Jet<T, N>& x) { return -x + s; }
template <typename T, int N> Jet<T, N> operator*(const Jet<T, N>& x, T s) { Jet<T, N> r; r.a = x.a * s; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * s; return r; }
template <typename T, int N> Jet<T, N> operator*(T s, const Jet<T, N>& x) { return x * s; }
template <typename T, int N> Jet<T, N> operator/(const Jet<T, N>& x, T s) { return x * (T(1) / s); }
template <typename T, int N> Jet<T, N> operator/(T s, const Jet<T, N>& x) { return Jet<T, N>(s) / x; }
template <typename T, int N> Jet<T, N>& operator+=(Jet<T, N>& x, const Jet<T, N>& y) { x = x + y; return x; }
template <typename T, int N> Jet<T, N>& operator-=(Jet<T, N>& x, const Jet<T, N>& y) { x = x - y; return x; }
template <typename T, int N> Jet<T, N>& operator*=(Jet<T, N>& x, const Jet<T, N>& y) { x = x * y; return x; }
template <typename T, int N> bool operator>(const Jet<T, N>& x, const Jet<T, N>& y) { return x.a > y.a; }
template <typename T, int N> bool operator<(const Jet<T, N>& x, const Jet<T, N>& y) { return x.a < y.a; }
template <typename T, int N> bool operator>(const Jet<T, N>& x, T s) { return x.a > s; }
template <typename T, int N> bool operator<(const Jet<T, N>& x, T s) { return x.a < s; }
template <typename T, int N> Jet<T, N> sqrt(const Jet<T, N>& x) { Jet<T, N> r; r.a = std::sqrt(x.a); const T d = T(1) / (T(2) * r.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
template <typename T, int N> Jet<T, N> sin(const Jet<T, N>& x) { Jet<T, N> r; r.a = std::sin(x.a); const T d = std::cos(x.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
template <typename T, int N> Jet<T, N> cos(const Jet<T, N>& x) { Jet<T, N> r; r.a = std::cos(x.a); const T d = -std::sin(x.a); for (int i = 0; i < N; ++i) r.v[i] = x.v[i] * d; return r; }
inline double sqrt(double x) { return std::sqrt(x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }

// ---- cost functions --------------------------------------------------------------------------
class CostFunction {
 public:
  virtual ~CostFunction() {}
  // parameters[b] = block b; jacobians[b] (may be null) = row-major num_residuals x block size
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  int num_residuals() const { return num_residuals_; }
  const std::vector<int>& parameter_block_sizes() const { return sizes_; }
 protected:
  int num_residuals_ = 0;
  std::vector<int> sizes_;
};

// AutoDiffCostFunction<Functor, kNumResiduals, N0, N1, N2, N3> — four parameter blocks, as
// src/BAConstraint.cpp:83 instantiates it (<BAConstraint, 2, 3, 3, 3, 6>).  Owns the functor.
template <typename Functor, int kNumResiduals, int N0, int N1, int N2, int N3>
class AutoDiffCostFunction : public CostFunction {
 public:
  explicit AutoDiffCostFunction(Functor* f) : f_(f) { num_residuals_ = kNumResiduals; sizes_ = {N0, N1, N2, N3}; }
  ~AutoDiffCostFunction() override { delete f_; }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    if (!jacobians) return (*f_)(p[0], p[1], p[2], p[3], residuals);
    constexpr int N = N0 + N1 + N2 + N3;
    typedef Jet<double, N> J;
    J x0[N0], x1[N1], x2[N2], x3[N3], r[kNumResiduals];
    int k = 0;
    for (int i = 0; i < N0; ++i) x0[i] = J(p[0][i], k++);
    for (int i = 0; i < N1; ++i) x1[i] = J(p[1][i], k++);
    for (int i = 0; i < N2; ++i) x2[i] = J(p[2][i], k++);
    for (int i = 0; i < N3; ++i) x3[i] = J(p[3][i], k++);
    if (!(*f_)(x0, x1, x2, x3, r)) return false;
    const int off[4] = {0, N0, N0 + N1, N0 + N1 + N2};
    const int sz[4] = {N0, N1, N2, N3};
    for (int i = 0; i < kNumResiduals; ++i) {
      residuals[i] = r[i].a;
      for (int b = 0; b < 4; ++b)
        if (jacobians[b])
          for (int j = 0; j < sz[b]; ++j) jacobians[b][i * sz[b] + j] = r[i].v[off[b] + j];
    }
    return true;
  }
 private:
  Functor* f_;
};

}  // namespace ceres

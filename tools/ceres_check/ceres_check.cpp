// ceres_check — the reference's solve (ceres::Solve as called at /root/reference/src/BundleAdjustment.cpp:108 with the options
// of app/main.cpp:21-28) on a BAL text file, with the 9-parameter BAL camera as an autodiff functor <2, 9, 3>, printing Ceres'
// per-iteration table as JSON lines so that tools/ceres_check/compare.py can lay it beside the HIP path's and the oracle's.
// Built only where find_package(Ceres) succeeds (CMakeLists.txt); it has never been compiled in the build container or on the
// GPU box, which have no Ceres.
//
// usage: ceres_check problem.txt [iterative|sparse_schur|dense_schur] [max_iterations] [threads]
#include <ceres/ceres.h>
#include <ceres/rotation.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

// residual = predicted - observed; p = -P.xy / P.z; distortion 1 + k1 |p|^2 + k2 |p|^4 (the BAL convention)
struct BalReprojection {
  BalReprojection(double x, double y) : ox(x), oy(y) {}
  template <typename T>
  bool operator()(const T* const cam, const T* const X, T* r) const {
    T P[3];
    ceres::AngleAxisRotatePoint(cam, X, P);
    P[0] += cam[3]; P[1] += cam[4]; P[2] += cam[5];
    const T px = -P[0] / P[2], py = -P[1] / P[2];
    const T n = px * px + py * py;
    const T d = T(1.0) + n * (cam[7] + cam[8] * n);
    r[0] = cam[6] * d * px - T(ox);
    r[1] = cam[6] * d * py - T(oy);
    return true;
  }
  double ox, oy;
};

struct Table : ceres::IterationCallback {
  ceres::CallbackReturnType operator()(const ceres::IterationSummary& s) override {
    std::printf("{\"iteration\": %d, \"cost\": %.17g, \"cost_change\": %.17g, \"gradient_max_norm\": %.17g, \"step_norm\": %.17g, "
                "\"relative_decrease\": %.17g, \"trust_region_radius\": %.17g, \"linear_solver_iterations\": %d, "
                "\"step_is_successful\": %d}\n",
                s.iteration, s.cost, s.cost_change, s.gradient_max_norm, s.step_norm, s.relative_decrease, s.trust_region_radius,
                s.linear_solver_iterations, s.step_is_successful ? 1 : 0);
    return ceres::SOLVER_CONTINUE;
  }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s problem.txt [iterative|sparse_schur|dense_schur] [max_iterations] [threads]\n", argv[0]); return 2; }
  const std::string solver = argc > 2 ? argv[2] : "iterative";
  const int max_it = argc > 3 ? std::atoi(argv[3]) : 80;
  const int threads = argc > 4 ? std::atoi(argv[4]) : 4;
  std::FILE* fh = std::fopen(argv[1], "r");
  if (!fh) { std::perror(argv[1]); return 1; }
  int n_cam = 0, n_pt = 0, n_obs = 0;
  if (std::fscanf(fh, "%d %d %d", &n_cam, &n_pt, &n_obs) != 3) { std::fprintf(stderr, "not a BAL file\n"); return 1; }
  std::vector<int> oc(n_obs), op(n_obs);
  std::vector<double> oxy(2 * (size_t)n_obs), cams(9 * (size_t)n_cam), pts(3 * (size_t)n_pt);
  for (int i = 0; i < n_obs; ++i)
    if (std::fscanf(fh, "%d %d %lf %lf", &oc[i], &op[i], &oxy[2 * i], &oxy[2 * i + 1]) != 4) { std::fprintf(stderr, "short file\n"); return 1; }
  // scalars are rounded through float as the reference loader parses them (src/BalDataloader.cpp:32-33,44-45,55-56)
  for (auto& v : oxy) v = (double)(float)v;
  for (auto& v : cams) { if (std::fscanf(fh, "%lf", &v) != 1) return 1; v = (double)(float)v; }
  for (auto& v : pts) { if (std::fscanf(fh, "%lf", &v) != 1) return 1; v = (double)(float)v; }
  std::fclose(fh);

  ceres::Problem problem;
  for (int i = 0; i < n_obs; ++i)
    problem.AddResidualBlock(new ceres::AutoDiffCostFunction<BalReprojection, 2, 9, 3>(new BalReprojection(oxy[2 * i], oxy[2 * i + 1])),
                             nullptr, &cams[9 * (size_t)oc[i]], &pts[3 * (size_t)op[i]]);
  ceres::Solver::Options o;
  o.minimizer_type = ceres::TRUST_REGION;
  o.trust_region_strategy_type = ceres::LEVENBERG_MARQUARDT;
  o.use_nonmonotonic_steps = false;
  o.max_num_iterations = max_it;
  o.num_threads = threads;
  if (solver == "iterative") { o.linear_solver_type = ceres::ITERATIVE_SCHUR; o.preconditioner_type = ceres::JACOBI; o.eta = 0.1; }
  else if (solver == "sparse_schur") o.linear_solver_type = ceres::SPARSE_SCHUR;
  else o.linear_solver_type = ceres::DENSE_SCHUR;
  Table table;
  o.callbacks.push_back(&table);
  ceres::Solver::Summary s;
  ceres::Solve(o, &problem, &s);
  std::printf("{\"final_cost\": %.17g, \"initial_cost\": %.17g, \"iterations\": %d, \"total_time_s\": %.6f, \"message\": \"%s\"}\n", s.final_cost,
              s.initial_cost, (int)s.iterations.size() - 1, s.total_time_in_seconds, s.message.c_str());
  return 0;
}

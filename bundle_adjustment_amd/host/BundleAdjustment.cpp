// BundleAdjustment behind the reference's class surface
// (/root/reference/src/BundleAdjustment.cpp:16-160).  createProblem() hands the observation
// lists to libmiba (filter + sort-by-point + tiling + upload happen there); solve() runs the
// HIP Levenberg-Marquardt / implicit-Schur PCG through the C ABI (include/miba.h) instead of
// ceres::Solve (:108).  No per-observation cost-function objects are created on this path.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <vector>

#include "bundleadjust/BundleAdjustment.h"
#include "bundleadjust/BAConstraint.h"
#include "MeshWriterCallback.h"
#include "miba.h"

namespace {

void check(int rc, miba_handle h, const char *what) {
    if (rc != MIBA_OK) throw std::runtime_error(std::string(what) + ": " + miba_last_error(h));
}

struct CallbackContext {
    miba_handle h;
    bool update_state;
    double *R, *T, *K, *X;
    std::vector<ceres::IterationCallback *> *callbacks;
    std::vector<ceres::IterationSummary> *log;
};

int iteration_trampoline(const miba_iteration *it, void *user) {
    CallbackContext *c = static_cast<CallbackContext *>(user);
    ceres::IterationSummary s;
    s.iteration = it->iteration;
    s.step_is_valid = it->step_is_valid != 0;
    s.step_is_successful = it->step_is_successful != 0;
    s.cost = it->cost;
    s.cost_change = it->cost_change;
    s.gradient_max_norm = it->gradient_max_norm;
    s.gradient_norm = it->gradient_norm;
    s.step_norm = it->step_norm;
    s.relative_decrease = it->relative_decrease;
    s.trust_region_radius = it->trust_region_radius;
    s.eta = it->eta;
    s.linear_solver_iterations = it->linear_solver_iterations;
    s.iteration_time_in_seconds = it->iteration_time_in_seconds;
    s.cumulative_time_in_seconds = it->cumulative_time_in_seconds;
    c->log->push_back(s);
    if (c->callbacks->empty()) return MIBA_SOLVER_CONTINUE;
    // update_state_every_iteration: the user arrays hold the current state when callbacks run
    if (c->update_state && miba_get_state(c->h, c->R, c->T, c->K, c->X) != MIBA_OK) return MIBA_SOLVER_ABORT;
    for (ceres::IterationCallback *cb : *c->callbacks) {
        const ceres::CallbackReturnType r = (*cb)(s);
        if (r == ceres::SOLVER_ABORT) return MIBA_SOLVER_ABORT;
        if (r == ceres::SOLVER_TERMINATE_SUCCESSFULLY) return MIBA_SOLVER_TERMINATE_SUCCESSFULLY;
    }
    return MIBA_SOLVER_CONTINUE;
}

}  // namespace

BundleAdjustment::BundleAdjustment(Dataloader *dataset, ceres::Solver::Options options)
        : dataset(dataset),
          options(options),
          problem(nullptr) {

    callback = new MeshWriterCallback(this, ceres::LinearSolverTypeToString(options.linear_solver_type));

    R = new double[dataset->getNumFrames() * 3];
    T = new double[dataset->getNumFrames() * 3];
    intrinsics = new double[dataset->getNumFrames() * 6];
    X = new double[dataset->getNumPoints() * 3]; // reconstructed 3D points

    dataset->initialize(R, T, intrinsics, X);
}

BundleAdjustment::~BundleAdjustment() {
    if (problem) miba_destroy(problem);
    delete[] R;
    delete[] T;
    delete[] X;
    delete[] intrinsics;

    delete callback;
}

void BundleAdjustment::createProblem() {
    std::cout << "Creating problem" << std::endl;
    auto observations = dataset->getObservations();
    const int n_obs = dataset->getNumObservations();

    // one (camera, point, x, y) record per observation, in the loader's order; invalid ones
    // (point == -1, NaN observation, NaN point) are dropped inside miba_set_problem exactly as
    // the reference's loop does (src/BundleAdjustment.cpp:49-60)
    std::vector<int32_t> obs_cam(n_obs), obs_pt(n_obs);
    std::vector<double> obs_xy(2 * (size_t) n_obs);
    int invalidObs = 0;
    for (int i = 0; i < n_obs; ++i) {
        obs_cam[i] = dataset->getObsCam(i);
        obs_pt[i] = dataset->getObsPoint(i);
        if (obs_pt[i] == -1) invalidObs++; // no 3d point to 2d point
        obs_xy[2 * (size_t) i] = observations[i].x;      // float -> double, as observation.cast<T>()
        obs_xy[2 * (size_t) i + 1] = observations[i].y;
    }

    if (!problem) {
        miba_config cfg{};
        cfg.device_id = -1;
        cfg.rank = 0;
        cfg.world_size = 1;
        miba_handle h = nullptr;
        check(miba_create(&cfg, &h), nullptr, "miba_create");
        problem = h;
    }
    miba_problem p{};
    p.n_cam = dataset->getNumFrames();
    p.n_pt = dataset->getNumPoints();
    p.n_obs = n_obs;
    p.R = R; p.T = T; p.K = intrinsics; p.X = X;
    p.obs_cam = obs_cam.data();
    p.obs_pt = obs_pt.data();
    p.obs_xy = obs_xy.data();
    check(miba_set_problem(problem, &p), problem, "miba_set_problem");

    // The reference's createProblem, in its checked-in debugging state, fixes the intrinsics and the rotation
    // of every observed camera (src/BundleAdjustment.cpp:77-79) and all of camera 0 (:88-90).  The north-star
    // configuration frees everything; BA_REFERENCE_CONSTANT_BLOCKS=1 reproduces the reference's marks.
    const char *rc = getenv("BA_REFERENCE_CONSTANT_BLOCKS");
    if (rc && rc[0] == '1') {
        std::vector<uint8_t> cc(3 * (size_t) p.n_cam, 0);
        for (int i = 0; i < n_obs; ++i) {
            const int c = obs_cam[i];
            if (obs_pt[i] == -1 || std::isnan(obs_xy[2 * (size_t) i]) || std::isnan(X[3 * (size_t) obs_pt[i]])) continue;
            cc[3 * (size_t) c + 0] = 1;   // rotation
            cc[3 * (size_t) c + 2] = 1;   // intrinsics
        }
        cc[0] = cc[1] = cc[2] = 1;        // fix 1st camera as reference
        check(miba_set_constant_blocks(problem, cc.data()), problem, "miba_set_constant_blocks");
    }

    std::cout << "Invalid observations: " << invalidObs << " out of " << dataset->getNumObservations() << std::endl;
    std::cout << "Creating problem end" << std::endl;
}

void BundleAdjustment::solve() {
    std::cout << "Solving problem" << std::endl;
    if (!problem) throw std::runtime_error("solve() called before createProblem()");
    ceres::IterationSummary sum;
    sum.iteration = -1;
    (*callback)(sum);

    options.update_state_every_iteration = true;
    options.callbacks.push_back(callback);

    miba_options o;
    miba_default_options(&o);
    o.max_num_iterations = options.max_num_iterations;
    o.linear_solver_type = (int) options.linear_solver_type;
    o.jacobi_scaling = options.jacobi_scaling ? 1 : 0;
    o.preconditioner_type = options.preconditioner_type == ceres::SCHUR_JACOBI ? MIBA_PRECOND_SCHUR_JACOBI : MIBA_PRECOND_JACOBI;
    o.max_linear_solver_iterations = options.max_linear_solver_iterations;
    o.min_linear_solver_iterations = options.min_linear_solver_iterations;
    o.max_num_consecutive_invalid_steps = options.max_num_consecutive_invalid_steps;
    o.update_state_every_iteration = options.update_state_every_iteration ? 1 : 0;
    o.minimizer_progress_to_stdout = options.minimizer_progress_to_stdout ? 1 : 0;
    o.initial_trust_region_radius = options.initial_trust_region_radius;
    o.max_trust_region_radius = options.max_trust_region_radius;
    o.min_trust_region_radius = options.min_trust_region_radius;
    o.min_relative_decrease = options.min_relative_decrease;
    o.min_lm_diagonal = options.min_lm_diagonal;
    o.max_lm_diagonal = options.max_lm_diagonal;
    o.function_tolerance = options.function_tolerance;
    o.gradient_tolerance = options.gradient_tolerance;
    o.parameter_tolerance = options.parameter_tolerance;
    o.eta = options.eta;
    o.max_solver_time_in_seconds = options.max_solver_time_in_seconds;

    // the state may have been edited through the getters since createProblem()
    check(miba_set_state(problem, R, T, intrinsics, X), problem, "miba_set_state");

    summary = ceres::Solver::Summary();
    CallbackContext ctx{problem, options.update_state_every_iteration, R, T, intrinsics, X, &options.callbacks,
                        &summary.iterations};
    miba_summary ms;
    check(miba_solve(problem, &o, iteration_trampoline, &ctx, &ms), problem, "miba_solve");
    check(miba_get_state(problem, R, T, intrinsics, X), problem, "miba_get_state");
    options.callbacks.pop_back();

    summary.termination_type = (ceres::TerminationType) ms.termination_type;
    summary.message = ms.message;
    summary.initial_cost = ms.initial_cost;
    summary.final_cost = ms.final_cost;
    summary.num_successful_steps = ms.num_successful_steps;
    summary.num_unsuccessful_steps = ms.num_unsuccessful_steps;
    summary.num_residual_blocks = ms.num_residual_blocks;
    summary.num_residuals = 2 * ms.num_residual_blocks;
    summary.num_parameter_blocks = ms.num_active_cameras + ms.num_active_points;
    summary.num_parameters = 9 * ms.num_active_cameras + 3 * ms.num_active_points;
    summary.num_linear_solver_iterations = ms.num_linear_solver_iterations;
    summary.total_time_in_seconds = ms.total_time_in_seconds;
    summary.linear_solver_time_in_seconds = ms.linear_solver_time_in_seconds;
    summary.jacobian_evaluation_time_in_seconds = ms.jacobian_time_in_seconds;
    summary.residual_evaluation_time_in_seconds = ms.residual_time_in_seconds;
    summary.linear_solver_type_given = options.linear_solver_type;
    summary.linear_solver_used = options.linear_solver_type == ceres::ITERATIVE_SCHUR
                                 ? (options.preconditioner_type == ceres::SCHUR_JACOBI
                                        ? "ITERATIVE_SCHUR (HIP implicit-Schur PCG, SCHUR_JACOBI preconditioner)"
                                        : "ITERATIVE_SCHUR (HIP implicit-Schur PCG, JACOBI preconditioner)")
                                 : ms.exact_linear_solver
                                       ? "DENSE_SCHUR (HIP explicit Schur complement + blocked Cholesky on the device)"
                                       : "ITERATIVE_SCHUR (HIP implicit-Schur PCG run to q-tolerance 1e-12 in place of the exact solver: "
                                         "more than 2048 cameras, or MIBA_DENSE=0 / MIBA_MF=0)";

    std::cout << ceres::LinearSolverTypeToString(options.linear_solver_type) << std::endl;
    std::cout << summary.FullReport() << std::endl;

    std::cout << "Solving problem end" << std::endl;
}

double *BundleAdjustment::getRotation(size_t cameraIndex) {
    return &R[cameraIndex * 3];
}

double *BundleAdjustment::getTranslation(size_t cameraIndex) {
    return &T[cameraIndex * 3];
}

double *BundleAdjustment::getIntrinsics(size_t pointIndex) {
    return &intrinsics[pointIndex * 6];
}

double *BundleAdjustment::getPoint(size_t pointIndex) {
    return &X[pointIndex * 3];
}

void BundleAdjustment::WriteToPLYFile(std::string filename) {
    // ASCII PLY: cameras (translation vector, green) then points, 4 decimals
    // (src/BundleAdjustment.cpp:131-160)
    FILE *of = fopen(filename.c_str(), "w");
    if (!of) throw std::runtime_error("cannot open " + filename);
    fprintf(of, "ply"
                "\nformat ascii 1.0"
                "\nelement vertex %d"
                "\nproperty float x"
                "\nproperty float y"
                "\nproperty float z"
                "\nproperty uchar red"
                "\nproperty uchar green"
                "\nproperty uchar blue"
                "\nend_header\n", this->dataset->getNumPoints() + this->dataset->getNumFrames());

    for (int i = 0; i < this->dataset->getNumFrames(); ++i) {
        auto cam = getTranslation(i);
        fprintf(of, "%.4f %.4f %.4f 0 255 0\n", cam[0], cam[1], cam[2]);
    }

    for (int i = 0; i < this->dataset->getNumPoints(); ++i) {
        auto point = getPoint(i);
        Eigen::Vector3i bgr = this->dataset->getPointColor(i);
        if (!std::isnan(point[0])) {
            fprintf(of, "%.4f %.4f %.4f %d %d %d\n", point[0], point[1], point[2], bgr(0), bgr(1), bgr(2));
        } else {
            fprintf(of, "0 0 0 %d %d %d\n", bgr(0), bgr(1), bgr(2));
        }
    }
    fclose(of);
}

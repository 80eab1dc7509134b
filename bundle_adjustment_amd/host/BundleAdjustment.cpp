// BundleAdjustment behind the reference's class surface
// (/root/reference/src/BundleAdjustment.cpp:16-160).  createProblem() hands the observation
// lists to libmiba (filter + sort-by-point + tiling + upload happen there); solve() runs the
// HIP Levenberg-Marquardt / implicit-Schur PCG through the C ABI (include/miba.h) instead of
// ceres::Solve (:108).  No per-observation cost-function objects are created on this path.
#include <charconv>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
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

// BA_DEBUG_TIMING=1: wall time of each public method on stderr
struct PhaseTimer {
    const char *what;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PhaseTimer(const char *w) : what(w) {}
    ~PhaseTimer() {
        if (getenv("BA_DEBUG_TIMING"))
            fprintf(stderr, "[BundleAdjustment] %-18s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};

}  // namespace

BundleAdjustment::BundleAdjustment(Dataloader *dataset, ceres::Solver::Options options)
        : dataset(dataset),
          options(options),
          problem(nullptr),
          plyQueue(nullptr) {

    PhaseTimer pt("constructor");
    callback = new MeshWriterCallback(this, ceres::LinearSolverTypeToString(options.linear_solver_type));

    R = new double[dataset->getNumFrames() * 3];
    T = new double[dataset->getNumFrames() * 3];
    intrinsics = new double[dataset->getNumFrames() * 6];
    X = new double[dataset->getNumPoints() * 3]; // reconstructed 3D points

    dataset->initialize(R, T, intrinsics, X);
}

static void ply_queue_close(PlyQueue *q);   // drains and deletes (defined with the queue, below)

BundleAdjustment::~BundleAdjustment() {
    ply_queue_close(plyQueue);
    if (problem) miba_destroy(problem);
    delete[] R;
    delete[] T;
    delete[] X;
    delete[] intrinsics;

    delete callback;
}

void BundleAdjustment::createProblem() {
    PhaseTimer pt("createProblem");
    std::cout << "Creating problem" << std::endl;
    auto observations = dataset->getObservations();
    const int n_obs = dataset->getNumObservations();

    // one (camera, point, x, y) record per observation, in the loader's order; invalid ones
    // (point == -1, NaN observation, NaN point) are dropped inside miba_set_problem exactly as
    // the reference's loop does (src/BundleAdjustment.cpp:49-60)
    std::vector<int32_t> obs_cam(n_obs), obs_pt(n_obs);
    std::vector<double> obs_xy(2 * (size_t) n_obs);
    int invalidObs = 0;
    {
        // ranges of observations converted by worker threads (the accessors are const and the ranges disjoint)
        unsigned n_thr = std::thread::hardware_concurrency();
        if (n_thr < 1) n_thr = 1;
        if (n_thr > 16) n_thr = 16;
        if (n_obs < (1 << 18)) n_thr = 1;
        std::vector<int> invalid(n_thr, 0);
        auto convert = [&](unsigned t) {
            const long long a = (long long) n_obs * t / n_thr, b = (long long) n_obs * (t + 1) / n_thr;
            for (long long i = a; i < b; ++i) {
                obs_cam[i] = dataset->getObsCam((int) i);
                obs_pt[i] = dataset->getObsPoint((int) i);
                if (obs_pt[i] == -1) invalid[t]++; // no 3d point to 2d point
                obs_xy[2 * (size_t) i] = observations[i].x;      // float -> double, as observation.cast<T>()
                obs_xy[2 * (size_t) i + 1] = observations[i].y;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n_thr; ++t) th.emplace_back(convert, t);
        convert(0);
        for (auto &x : th) x.join();
        for (int v : invalid) invalidObs += v;
    }

    if (!problem) {
        // One BundleAdjustment object = one handle, whatever the number of GPUs (SURVEY 8b "Threading"): the reference's
        // call site (app/main.cpp:30-33) has nowhere to say "8 GPUs", so the count comes from the environment.
        // BA_NUM_GPUS=N (1..8): devices 0..N-1 of this process, points sharded over them; BA_DEVICES=0,0 names the
        // ordinals explicitly (a repeated ordinal puts several shards on one GPU).
        std::vector<int32_t> devs;
        if (const char *dl = getenv("BA_DEVICES")) {
            for (const char *q = dl; *q;) {
                char *endp = nullptr;
                const long v = strtol(q, &endp, 10);
                if (endp == q) break;
                devs.push_back((int32_t) v);
                q = (*endp == ',') ? endp + 1 : endp;
            }
        } else if (const char *ng = getenv("BA_NUM_GPUS")) {
            for (int i = 0; i < atoi(ng); ++i) devs.push_back(i);
        }
        miba_handle h = nullptr;
        if (devs.size() > 1) {
            check(miba_create_multi(devs.data(), (int32_t) devs.size(), 0, &h), nullptr, "miba_create_multi");
        } else {
            miba_config cfg{};
            cfg.device_id = devs.empty() ? -1 : devs[0];
            cfg.rank = 0;
            cfg.world_size = 1;
            check(miba_create(&cfg, &h), nullptr, "miba_create");
        }
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
    // BA_CAMERA_MODEL=reference: the cost functor the reference has checked in (src/BAConstraint.cpp:15-80) instead of the BAL
    // camera; with BA_REFERENCE_CONSTANT_BLOCKS=1 this is createProblem as it stands in the reference (:67-90)
    const char *cmod = getenv("BA_CAMERA_MODEL");
    const bool ref_model = cmod && std::string(cmod) == "reference";
    p.camera_model = ref_model ? MIBA_MODEL_REFERENCE : MIBA_MODEL_BAL;
    BAConstraint::useReferenceModel(ref_model);
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
    PhaseTimer pt("solve");
    std::cout << "Solving problem" << std::endl;
    if (!problem) throw std::runtime_error("solve() called before createProblem()");
    ceres::IterationSummary sum;
    sum.iteration = -1;
    (*callback)(sum);

    options.update_state_every_iteration = true;
    const bool mesh_pushed = static_cast<MeshWriterCallback *>(callback)->isEnabled();
    if (mesh_pushed) options.callbacks.push_back(callback);

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
    // update_state_every_iteration + callbacks (src/BundleAdjustment.cpp:104-105) off the solver's critical path: the library
    // copies the state of every iteration into pinned snapshots on a second stream and runs the trampoline below on a worker
    // thread, from the snapshot, while the device works on the next iteration (BA_ASYNC_CALLBACKS=0: on this thread, between
    // iterations, as ceres::Solve does).  The arrays R, T, intrinsics, X are only touched by that worker during solve().
    {
        const char *ac = getenv("BA_ASYNC_CALLBACKS");
        o.async_callbacks = (ac && ac[0] == '0') ? 0 : 1;
    }
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
    // Loss function of every residual block: the reference passes nullptr to AddResidualBlock (src/BundleAdjustment.cpp:69),
    // which stays the default.  BA_LOSS=huber:<a> or cauchy:<a> applies ceres::HuberLoss(a) / ceres::CauchyLoss(a) instead
    // (a in pixels), e.g. for BAL files with mismatched tracks.
    if (const char *ls = getenv("BA_LOSS")) {
        const std::string spec(ls);
        const size_t colon = spec.find(':');
        const std::string kind = spec.substr(0, colon);
        if (kind == "huber") o.loss_type = MIBA_LOSS_HUBER;
        else if (kind == "cauchy") o.loss_type = MIBA_LOSS_CAUCHY;
        else if (!kind.empty() && kind != "none") throw std::invalid_argument("BA_LOSS: expected huber:<a>, cauchy:<a> or none, got " + spec);
        if (o.loss_type != MIBA_LOSS_NONE) o.loss_scale = colon == std::string::npos ? 1.0 : std::atof(spec.c_str() + colon + 1);
    }

    // the state may have been edited through the getters since createProblem()
    check(miba_set_state(problem, R, T, intrinsics, X), problem, "miba_set_state");

    summary = ceres::Solver::Summary();
    CallbackContext ctx{problem, options.update_state_every_iteration, R, T, intrinsics, X, &options.callbacks,
                        &summary.iterations};
    miba_summary ms;
    check(miba_solve(problem, &o, iteration_trampoline, &ctx, &ms), problem, "miba_solve");
    check(miba_get_state(problem, R, T, intrinsics, X), problem, "miba_get_state");
    if (mesh_pushed) options.callbacks.pop_back();
    flushPlyWrites();   // the per-iteration dumps queued by the mesh callback are on disk when solve() returns

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
                                       ? "DENSE_SCHUR / SPARSE_SCHUR (HIP explicit Schur complement + blocked Cholesky on the device over 64 x 64 "
                                         "tiles of the sparse Schur complement; dense up to 2048 cameras where the tiles are not available)"
                                       : "ITERATIVE_SCHUR (HIP implicit-Schur PCG run to q-tolerance 1e-12 in place of the exact solver: "
                                         "the tiles of the Schur complement do not fit the free device memory, or MIBA_DENSE=0)";

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

namespace {

// ASCII PLY: cameras (translation vector, green) then points, 4 decimals (src/BundleAdjustment.cpp:131-160).
// Same text as fprintf("%.4f %.4f %.4f %d %d %d\n") line by line (std::to_chars in fixed notation rounds the way
// printf does), formatted by worker threads into per-range buffers and written in order: a Final-13682 dump is
// 4.5 M lines, and the reference writes one per LM iteration.
void write_ply(const std::string &filename, int n_frames, int n_points, const double *T, const double *X, const Dataloader *dataset) {
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
                "\nend_header\n", n_points + n_frames);
    const long long n_lines = (long long) n_frames + n_points;
    unsigned n_thr = std::thread::hardware_concurrency();
    if (n_thr < 1) n_thr = 1;
    if (n_thr > 16) n_thr = 16;
    if (n_lines < (1 << 16)) n_thr = 1;
    std::vector<std::string> chunk(n_thr);
    auto put_fixed4 = [](std::string &out, double v) {
        char tmp[400];   // %.4f of the largest double has 309 integer digits
        auto r = std::to_chars(tmp, tmp + sizeof(tmp), v, std::chars_format::fixed, 4);
        out.append(tmp, r.ptr);
    };
    auto put_int = [](std::string &out, int v) {
        char tmp[16];
        auto r = std::to_chars(tmp, tmp + sizeof(tmp), v);
        out.append(tmp, r.ptr);
    };
    auto work = [&](unsigned t) {
        std::string &out = chunk[t];
        const long long a = n_lines * t / n_thr, b = n_lines * (t + 1) / n_thr;
        out.reserve((size_t) (b - a) * 40);
        for (long long l = a; l < b; ++l) {
            if (l < n_frames) {
                const double *cam = T + 3 * (size_t) l;
                put_fixed4(out, cam[0]); out += ' '; put_fixed4(out, cam[1]); out += ' '; put_fixed4(out, cam[2]);
                out += " 0 255 0\n";
            } else {
                const int i = (int) (l - n_frames);
                const double *point = X + 3 * (size_t) i;
                Eigen::Vector3i bgr = dataset->getPointColor(i);
                if (!std::isnan(point[0])) {
                    put_fixed4(out, point[0]); out += ' '; put_fixed4(out, point[1]); out += ' '; put_fixed4(out, point[2]);
                } else {
                    out += "0 0 0";
                }
                out += ' '; put_int(out, bgr(0)); out += ' '; put_int(out, bgr(1)); out += ' '; put_int(out, bgr(2)); out += '\n';
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n_thr; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    for (unsigned t = 0; t < n_thr; ++t) fwrite(chunk[t].data(), 1, chunk[t].size(), of);
    fclose(of);
}

}  // namespace

// Per-iteration dumps off the solver's critical path: the mesh callback hands a snapshot of the
// translations and points to one writer thread and the LM loop goes on; at most two snapshots wait (a third caller
// blocks), files are written in order, and solve() / the destructor wait for the queue to drain.
struct PlyQueue {
    struct Job { std::string name; std::vector<double> T, X; };
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> jobs;
    bool busy = false, stop = false;
    std::string error;
    int n_frames, n_points;
    const Dataloader *dataset;
    std::thread worker;
    PlyQueue(int nf, int np, const Dataloader *d) : n_frames(nf), n_points(np), dataset(d), worker([this] { run(); }) {}
    ~PlyQueue() {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        worker.join();
    }
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || !jobs.empty(); });
                if (jobs.empty()) return;
                j = std::move(jobs.front());
                jobs.pop_front();
                busy = true;
            }
            cv.notify_all();
            try { write_ply(j.name, n_frames, n_points, j.T.data(), j.X.data(), dataset); }
            catch (const std::exception &e) { std::lock_guard<std::mutex> l(m); if (error.empty()) error = e.what(); }
            { std::lock_guard<std::mutex> l(m); busy = false; }
            cv.notify_all();
        }
    }
    void push(Job &&j) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return jobs.size() < 2; });
        jobs.push_back(std::move(j));
        l.unlock();
        cv.notify_all();
    }
    void drain() {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return jobs.empty() && !busy; });
        if (!error.empty()) { std::string e = error; error.clear(); throw std::runtime_error(e); }
    }
};

static void ply_queue_close(PlyQueue *q) {
    if (!q) return;
    try { q->drain(); } catch (const std::exception &e) { fprintf(stderr, "PLY writer: %s\n", e.what()); }
    delete q;
}

void BundleAdjustment::WriteToPLYFile(std::string filename) {
    PhaseTimer pt("WriteToPLYFile");
    flushPlyWrites();   // keep files in the order they were asked for
    write_ply(filename, this->dataset->getNumFrames(), this->dataset->getNumPoints(), T, X, this->dataset);
}

void BundleAdjustment::WriteToPLYFileAsync(std::string filename) {
    const char *e = getenv("BA_ASYNC_PLY");
    if (e && e[0] == '0') { WriteToPLYFile(filename); return; }
    const int nf = this->dataset->getNumFrames(), np = this->dataset->getNumPoints();
    if (!plyQueue) plyQueue = new PlyQueue(nf, np, this->dataset);
    PlyQueue::Job j;
    j.name = std::move(filename);
    j.T.assign(T, T + 3 * (size_t) nf);
    j.X.assign(X, X + 3 * (size_t) np);
    plyQueue->push(std::move(j));
}

void BundleAdjustment::flushPlyWrites() {
    if (plyQueue) plyQueue->drain();
}

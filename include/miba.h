/*
 * miba.h — C ABI of libmiba: the MI355X-native (gfx950, HIP) Levenberg-Marquardt /
 * implicit-Schur-PCG bundle-adjustment solver that replaces the Ceres-backed inner loop of
 * Dawars/bundle_adjustment behind its BundleAdjustment / BAConstraint C++ API.
 *
 * Plain pointers and sizes only; no C++/torch/HIP types.  Every entry point returns an int
 * status (MIBA_OK = 0, < 0 error); miba_last_error() gives the text.  The library copies
 * caller data in at miba_set_problem()/miba_set_state() and out at miba_get_state(); it
 * never keeps caller pointers.  A handle is not thread-safe.  A handle of miba_create drives ONE GPU (multi-GPU: one
 * process and handle per GPU, joined by an RCCL communicator); a handle of miba_create_multi drives several GPUs from
 * one process.
 *
 * There is NO CPU fallback: every compute entry point fails with MIBA_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 *
 * Reference interfaces replaced (paths under /root/reference):
 *   miba_set_problem  <- BundleAdjustment::createProblem            src/BundleAdjustment.cpp:39-96
 *                        (filter :49-60, one residual block per observation :67-74,
 *                         BAConstraint::create src/BAConstraint.cpp:82-86)
 *   miba_set_state    <- Dataloader::initialize into R,T,intrinsics,X   src/BundleAdjustment.cpp:22-27
 *   miba_solve        <- ceres::Solve(options,&problem,&summary)    src/BundleAdjustment.cpp:98-113
 *   miba_get_state    <- update_state_every_iteration + getRotation/getTranslation/
 *                        getIntrinsics/getPoint                     src/BundleAdjustment.cpp:104,115-129
 *   miba_evaluate     <- BAConstraint::operator()<double / Jet>     src/BAConstraint.cpp:15-80
 *                        (as called by ceres::Problem::Evaluate)
 *   miba_options      <- ceres::Solver::Options fields used at      app/main.cpp:21-28,
 *                                                                   src/BundleAdjustment.cpp:104-105
 *   miba_iteration    <- ceres::IterationSummary (callback argument) src/MeshWriterCallback.h:17-23
 *   miba_summary      <- ceres::Solver::Summary                     src/BundleAdjustment.cpp:107-110
 */
#ifndef MIBA_H
#define MIBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct miba_handle_s* miba_handle;

enum {
  MIBA_OK = 0,
  MIBA_ERR_INVALID = -1,    /* bad argument / call order */
  MIBA_ERR_NO_DEVICE = -2,  /* no usable HIP device (there is no CPU fallback) */
  MIBA_ERR_HIP = -3,        /* a HIP runtime call failed */
  MIBA_ERR_RCCL = -4,       /* an RCCL call failed */
  MIBA_ERR_NOMEM = -5,
  MIBA_ERR_LIMIT = -6       /* problem exceeds a documented limit (e.g. n_cam >= 2^24) */
};

/* mirrors ceres::LinearSolverType as far as app/main.cpp:19 / BundleAdjustment.cpp:20 use it */
enum {
  MIBA_DENSE_NORMAL_CHOLESKY = 0,
  MIBA_DENSE_QR = 1,
  MIBA_SPARSE_NORMAL_CHOLESKY = 2,
  MIBA_DENSE_SCHUR = 3,
  MIBA_SPARSE_SCHUR = 4,
  MIBA_ITERATIVE_SCHUR = 5, /* device implicit-Schur PCG: the north-star path */
  MIBA_CGNR = 6
};

/* mirrors ceres::TerminationType */
enum { MIBA_CONVERGENCE = 0, MIBA_NO_CONVERGENCE = 1, MIBA_FAILURE = 2, MIBA_USER_SUCCESS = 3, MIBA_USER_FAILURE = 4 };

/* mirrors ceres::CallbackReturnType (src/MeshWriterCallback.h:22) */
enum { MIBA_SOLVER_CONTINUE = 0, MIBA_SOLVER_ABORT = 1, MIBA_SOLVER_TERMINATE_SUCCESSFULLY = 2 };

typedef struct {
  int32_t device_id;   /* HIP device ordinal, -1 = current device */
  int32_t rank;        /* shard owned by this process, 0 <= rank < world_size */
  int32_t world_size;  /* number of point shards (= processes = GPUs) */
  int32_t log_level;   /* 0 silent, 1 summary, 2 per-iteration */
  /* 128-byte RCCL unique id from miba_comm_unique_id() on rank 0, broadcast by the caller
   * (e.g. torch.distributed); NULL when world_size == 1. */
  const void* comm_unique_id;
} miba_config;

typedef struct {
  int32_t n_cam;
  int32_t n_pt;
  int64_t n_obs;
  /* parameter arrays in the reference layout (src/BundleAdjustment.cpp:22-25):
   * R[3*n_cam] angle-axis, T[3*n_cam], K[6*n_cam] = fx,fy,ox,oy,k1,k2 (BAL: fx=fy=f, ox=oy=0;
   * src/BalDataloader.cpp:143-148), X[3*n_pt]. */
  const double* R;
  const double* T;
  const double* K;
  const double* X;
  /* observations in the caller's (file) order; obs_pt == -1, NaN observation or NaN point
   * are skipped exactly as src/BundleAdjustment.cpp:49-60 does. obs_xy = x0,y0,x1,y1,...
   * (float-rounded values widened to double). */
  const int32_t* obs_cam;
  const int32_t* obs_pt;
  const double* obs_xy;
  /* MIBA_MODEL_BAL (0, the default): the 9-parameter BAL camera of the north star (rotate, translate, -P.xy/P.z, radial
   * distortion, f; residual predicted - observed).
   * MIBA_MODEL_REFERENCE (1): the reference's cost functor as checked in, BAConstraint::operator() src/BAConstraint.cpp:15-80:
   * P = [R(rot) | tr]^-1 [X; 1], projection (fx P.x/P.z + ox, fy P.y/P.z + oy), no distortion, residual observed - projected.
   * Its intrinsics block (fx, fy, ox, oy, k1, k2) is always constant, as the reference marks it (src/BundleAdjustment.cpp:77);
   * rotation and translation blocks are free unless miba_set_constant_blocks says otherwise.  Needs fx == fy (what the
   * reference's own data has: src/VirtualSensor.cpp:37-39); k1, k2 are ignored, as the functor ignores them.  miba_evaluate
   * returns the functor's residuals and the Jacobian columns rot, tr and three zero columns. */
  int32_t camera_model;
  int32_t reserved;
} miba_problem;
enum { MIBA_MODEL_BAL = 0, MIBA_MODEL_REFERENCE = 1 };

/* preconditioner of the implicit-Schur PCG: block diagonal of the camera part B = F'F + D^2 (Ceres' JACOBI,
 * the default the reference runs with) or of the Schur complement itself (Ceres' SCHUR_JACOBI) */
enum { MIBA_PRECOND_JACOBI = 0, MIBA_PRECOND_SCHUR_JACOBI = 1 };

typedef struct {
  int32_t max_num_iterations;               /* app/main.cpp:27 -> 80; Ceres default 50 */
  int32_t linear_solver_type;               /* MIBA_* above; all but ITERATIVE_SCHUR are solved exactly on the device
                                               (tiles of the Schur complement) or as tight-tolerance PCG (eta_exact), and reported so */
  int32_t jacobi_scaling;                   /* 1 */
  int32_t max_linear_solver_iterations;     /* 500 */
  int32_t min_linear_solver_iterations;     /* 0 */
  int32_t max_num_consecutive_invalid_steps;/* 5 */
  int32_t update_state_every_iteration;     /* src/BundleAdjustment.cpp:104 */
  int32_t minimizer_progress_to_stdout;     /* app/main.cpp:26 */
  int32_t pcg_check_every;                  /* host looks at the device PCG status every N its (>=1, default 4); the next iteration
                                               is enqueued before the host waits, so a look does not stall the stream; the
                                               device decides termination itself, launches enqueued past it return at once */
  int32_t preconditioner_type;              /* MIBA_PRECOND_*: ceres::Solver::Options::preconditioner_type for
                                               ITERATIVE_SCHUR; the reference leaves Ceres' default, JACOBI */
  double initial_trust_region_radius;       /* 1e4 */
  double max_trust_region_radius;           /* 1e16 */
  double min_trust_region_radius;           /* 1e-32 */
  double min_relative_decrease;             /* 1e-3 */
  double min_lm_diagonal;                   /* 1e-6 */
  double max_lm_diagonal;                   /* 1e32 */
  double function_tolerance;                /* 1e-6 */
  double gradient_tolerance;                /* 1e-10 */
  double parameter_tolerance;               /* 1e-8 */
  double eta;                               /* 1e-1 */
  double eta_exact;                         /* q-tolerance used for the non-iterative solver types (1e-12) */
  double max_solver_time_in_seconds;        /* 1e9 */
  /* robust loss applied to every residual block: the LossFunction argument of Problem::AddResidualBlock, for which
   * the reference passes nullptr (src/BundleAdjustment.cpp:69) = MIBA_LOSS_NONE, the default.  HUBER / CAUCHY are
   * ceres::HuberLoss(loss_scale) / ceres::CauchyLoss(loss_scale); every linear solver type and preconditioner. */
  int32_t loss_type;
  int32_t async_callbacks;                  /* 0 (default): the iteration callback runs on the calling thread between iterations, as
                                               ceres::IterationCallback does.  1: it runs on a worker thread of the solve while the solver
                                               works on the next iteration; after every iteration that changed the state the cameras and
                                               points are copied on a second stream into one of two pinned buffers, and miba_get_state
                                               called from inside the callback reads that snapshot (with one process per GPU: this rank's
                                               points only).  A stop verdict takes effect one iteration late. */
  double loss_scale;                        /* the 'a' of HuberLoss / CauchyLoss, in residual units (pixels) */
} miba_options;

enum { MIBA_LOSS_NONE = 0, MIBA_LOSS_HUBER = 1, MIBA_LOSS_CAUCHY = 2 };

typedef struct {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t linear_solver_iterations;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  double eta;
  double iteration_time_in_seconds;
  double cumulative_time_in_seconds;
  double model_cost_change;
} miba_iteration;

typedef struct {
  int32_t termination_type;
  int32_t num_iterations;           /* LM iterations run (successful + unsuccessful), iteration 0 excluded */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_linear_solver_iterations; /* total PCG iterations */
  int32_t num_residual_blocks;      /* observations that survived the filter (all shards) */
  int32_t num_active_cameras;
  int32_t num_active_points;
  double initial_cost;
  double final_cost;
  double total_time_in_seconds;     /* inside miba_solve, excluding callbacks */
  double linear_solver_time_in_seconds;
  double jacobian_time_in_seconds;
  double residual_time_in_seconds;
  char message[128];
  int32_t exact_linear_solver;      /* 1: the reduced system was factored on the device (dense Schur + Cholesky); 0: PCG */
  int32_t reserved;
} miba_summary;

/* per-iteration callback: return one of MIBA_SOLVER_*.  When
 * options.update_state_every_iteration != 0 the current state can be read with
 * miba_get_state() from inside the callback. */
typedef int (*miba_iteration_callback)(const miba_iteration* it, void* user);

/* Kernel ids for miba_get_kernel_times */
enum {
  MIBA_K_EVAL_JAC = 0,   /* K1  residual + Jacobian              */
  MIBA_K_EVAL_COST = 1,  /* K1' residual-only cost               */
  MIBA_K_SCHUR_INIT = 2, /* K2  per-point Schur init + camera block sums */
  MIBA_K_SCHUR_MATVEC = 3, /* K3 implicit Schur mat-vec          */
  MIBA_K_BACKSUB = 4,    /* K4  back-substitution + model cost change */
  MIBA_K_SCHUR_MATVEC_SKIPPED = 5, /* K3 launches enqueued ahead of the host poll that found the PCG finished and returned at once */
  MIBA_K_SCHUR_INIT_CAM = 6, /* K2b camera side of the Schur init (k_mf_camrows); MIBA_K_SCHUR_INIT is then the point side K2a */
  MIBA_K_CAM_REDUCE = 7,   /* camera-side sum of a product (k_cam_reduce_mf), sampled like K3 */
  MIBA_K_COUNT = 8
};

typedef struct {
  int64_t launches[MIBA_K_COUNT];
  double total_ms[MIBA_K_COUNT];   /* HIP-event time on the launch stream; inside the PCG the product is timed in one
                                      iteration out of 8 (MIBA_TIME_EVERY), launches[] counts the timed launches */
  int64_t n_slots;                 /* padded observation slots on this shard */
  int64_t n_obs_local;             /* real observations on this shard */
  int64_t n_pt_local;
  int64_t n_tiles;
  int32_t n_cam;
  int32_t product_layout;  /* 1: the implicit-Schur product walks camera-major sub-tiles (csrc/miba_mf2.hpp), 0: point-major wavelets */
  double product_clock_mhz; /* shader clock the last product launch ran at: (s_memtime ticks) / (s_memrealtime ticks) x 100 MHz over the
                               lifetime of one workgroup in the middle of its grid; 0 when not measured (point-major product) */
  int32_t pcg_launches;     /* kernel launches per ordinary PCG iteration of the last solve (product + tail = 2; with an all-reduce between
                               phase A and the rest of the tail 3 + the collective; without grid barriers one more) */
  int32_t pcg_exchange;     /* how q = S p crossed the ranks in the last solve: 0 one rank, 1 an all-reduce call between launches (RCCL,
                               the peer kernel or the host transport), 2 inside the tail kernel (peer transport, ranks of one process) */
} miba_kernel_times;

void miba_default_options(miba_options* o);

int miba_create(const miba_config* cfg, miba_handle* out);
/* One handle that drives n_devices GPUs (1..8) from ONE process: what the reference's single `BundleAdjustment ba{data,
 * options}` (app/main.cpp:30-33) needs to use a whole node.  Points are sharded over the devices exactly as over the
 * ranks of a one-process-per-GPU world (same plan, same kernels); every entry point takes the handle and runs one host
 * thread per device inside, the calling thread among them; callbacks run on the calling thread.  device_ids NULL = devices
 * 0..n-1.  The all-reduce transport is RCCL (ncclCommInitAll) by default; MIBA_ALLREDUCE=peer selects the library's own
 * 2-hop kernel over peer-mapped staging buffers, MIBA_ALLREDUCE=host a transport staged through host memory, which is
 * also the default when a device id repeats (several ranks on one GPU: rehearsals and tests; RCCL refuses that). */
int miba_create_multi(const int32_t* device_ids, int32_t n_devices, int32_t log_level, miba_handle* out);
int miba_destroy(miba_handle h);
/* text of the last error on this handle (h may be NULL: last error of a failed miba_create) */
const char* miba_last_error(miba_handle h);
/* 128 bytes; call on rank 0 only, then broadcast */
int miba_comm_unique_id(void* out128);

int miba_set_problem(miba_handle h, const miba_problem* p);
int miba_set_state(miba_handle h, const double* R, const double* T, const double* K, const double* X);

/* Constant parameter blocks: replaces ceres::Problem::SetParameterBlockConstant as called at
 * src/BundleAdjustment.cpp:77-79 (intrinsics and rotation of every observed camera) and :88-90 (camera 0).
 * cam_const[3*c + {0,1,2}] != 0 marks the rotation / translation / intrinsics (f,k1,k2) block of camera c
 * constant: its columns leave the problem (no step, no contribution to the gradient, step and parameter
 * norms), as in Ceres' reduced program.  NULL frees every block (the default after miba_set_problem).
 * Point blocks are always free, as in the reference. */
int miba_set_constant_blocks(miba_handle h, const uint8_t* cam_const);
int miba_get_state(miba_handle h, double* R, double* T, double* K, double* X);

/* Cost, and optionally residuals (2*n_obs), camera Jacobian blocks (18*n_obs, 2x9 row-major,
 * columns w0 w1 w2 t0 t1 t2 f k1 k2), point Jacobian blocks (6*n_obs, 2x3 row-major), all in
 * the CALLER'S observation order (skipped observations are zero-filled), and the gradient
 * J'r for cameras (9*n_cam) and points (3*n_pt).  Unscaled Jacobian, r = predicted - observed.
 * Any output pointer may be NULL.  With world_size > 1 each rank fills only its shard's
 * observations/points; cost and camera gradient are all-reduced. */
int miba_evaluate(miba_handle h, double* cost, double* residuals, double* jac_cam, double* jac_pt,
                  double* grad_cam, double* grad_pt);

int miba_solve(miba_handle h, const miba_options* opt, miba_iteration_callback cb, void* user,
               miba_summary* summary);
/* iteration records of the last miba_solve (iteration 0 first); *n = records available */
int miba_get_iterations(miba_handle h, miba_iteration* out, int32_t max, int32_t* n);

/* One damped linear solve at the current state (kernel-level parity hook): Jacobi scale at
 * the current state (if options.jacobi_scaling), D = sqrt(clamp(diag J'J)/radius), implicit
 * Schur PCG, back-substitution.  Outputs (any may be NULL): y_cam (9*n_cam) and y_pt (3*n_pt)
 * = solution of (J'J + D'D) y = J'r in scaled coordinates (before the LM negation), the
 * reduced right-hand side rhs (9*n_cam), S*probe for an optional probe vector (9*n_cam),
 * PCG iteration count and model cost change. */
int miba_linear_step(miba_handle h, const miba_options* opt, double radius, double* y_cam,
                     double* y_pt, double* rhs, const double* probe, double* s_probe,
                     int32_t* n_cg, double* model_cost_change);

/* Exact reduced solve.  A non-iterative options.linear_solver_type (what the reference app
 * asks for at app/main.cpp:19 and hands to ceres::Solve at src/BundleAdjustment.cpp:20,109) is served by an
 * explicit dense Schur complement and a blocked Cholesky on the device when 9*n_cam <= 18432 (and the
 * matrix-free kernels are active); larger problems fall back to PCG run to options.eta_exact.
 * MIBA_DENSE=0 in the environment forces that fallback.
 *
 * miba_get_reduced_system: the matrix S (9*n_cam x 9*n_cam, row-major, symmetric, Jacobi-scaled, LM diagonal
 * included) of the state prepared by the last miba_linear_step: parity hook for the assembly kernels.
 * miba_dense_spd_solve: the factorisation alone on a caller-supplied SPD matrix (row-major n x n, lower
 * triangle read), x = A^-1 b; *status = 0, or 2 when a pivot was not positive. */
int miba_get_reduced_system(miba_handle h, double* S_out);
int miba_dense_spd_solve(miba_handle h, int32_t n, const double* A, const double* b, double* x, int32_t* status);

/* miba_dense_spd_solve_static: NO COUNTERPART IN THE REFERENCE.  The same factorisation as miba_dense_spd_solve for
 * callers that cannot allocate device memory: tight serving loops, harnesses that forbid hipMalloc, the set-up phase
 * around a HIP graph capture (NOT inside one: the work runs on the handle's own stream and the call synchronises it, so
 * nothing of it would be captured).  It calls no hipMalloc / hipFree / allocating routine: the padded matrix, the diagonal-block
 * factors and the pivot flag live in a file-scope static __device__ array of the library (9.4 MB, held by every GPU
 * the library runs on until the process ends), whose device address is resolved once per device at miba_create.
 * All four pointers are DEVICE pointers on the handle's GPU: A (n x n row-major, lower triangle read), b (n),
 * x (n, output), status (one int32, output: 0, or 2 when a pivot was not positive).  n <= MIBA_STATIC_SOLVE_MAX_N,
 * larger sizes are refused with MIBA_ERR_LIMIT.  The work runs on the handle's stream and has finished on return.
 * Every call on one GPU shares that GPU's array: calls that overlap on one GPU (two handles, two threads) race. */
#define MIBA_STATIC_SOLVE_MAX_N 1024
int miba_dense_spd_solve_static(miba_handle h, int32_t n, const double* d_A, const double* d_b, double* d_x, int32_t* d_status);

/* Measurement hook: average time of `reps` back-to-back implicit-Schur products S*probe (K3 plus its
 * camera reduction) at the state prepared by the last miba_linear_step. probe has 9*n_cam entries. */
int miba_time_matvec(miba_handle h, const double* probe, int32_t reps, double* avg_ms);

int miba_get_kernel_times(miba_handle h, miba_kernel_times* out);
int miba_reset_kernel_times(miba_handle h);

/* Host-only planning helper (needs no GPU): how set_problem would tile and shard a problem.
 * tiles_per_rank / obs_per_rank have world_size entries. */
int miba_plan_shards(const miba_problem* p, int32_t world_size, int64_t* n_valid_obs,
                     int64_t* n_tiles, int64_t* n_slots, int64_t* tiles_per_rank,
                     int64_t* obs_per_rank);

/* Host-only: rank that owns each of the caller's points under the same plan (-1 = point has no
 * valid observation and is not part of the problem). owner has n_pt entries. */
int miba_plan_point_owner(const miba_problem* p, int32_t world_size, int32_t* owner);

/* Host-only (needs no GPU): the elimination order the exact solve's symbolic phase gives a tile graph (csrc/miba_plan.hpp,
 * order_tile_graph: index cuts into n_arcs arcs, the higher end of every crossing edge into the separator, minimum degree inside,
 * separator last); no counterpart in the reference (Ceres' ordering code plays this role behind ceres::Solve).  The graph has
 * n_rows vertices (tile rows) and n_edges undirected edges (eu[i], ev[i]); n_arcs is 1..64.  Outputs: pos[n_rows] = elimination step
 * of every row, arc_start[n_arcs + 1] = first step of every arc and of the separator rows (entries past *n_arcs_out repeat the
 * separator's start), the number of arcs the order ended up with (1: not dissected), the number of tiles below the diagonal after
 * fill, (cost_us, may be null) the modelled time of the numeric phase by which the symbolic phase chooses the number of arcs, and
 * (group_start[2 n_arcs + 1], group_level[2 n_arcs], n_groups; may be null) the groups of the order: the arcs (level 0), then the
 * separator sets by level; the groups of a level are factored side by side.  nested = 0: one separator set (level 1).  Test hook. */
int miba_plan_tile_order(int32_t n_rows, const int32_t* eu, const int32_t* ev, int64_t n_edges, int32_t n_arcs,
                         int32_t* pos, int32_t* arc_start, int32_t* n_arcs_out, int64_t* fill_tiles, double* cost_us,
                         int32_t nested, int32_t* group_start, int32_t* group_level, int32_t* n_groups);

/* Host-only diagnostic of the product kernel's layout (camera-major sub-tiles, csrc/miba_mf2.hpp) for one shard; no counterpart
 * in the reference (Ceres keeps its block-sparse Jacobian in one layout). out10 = { layout is a permutation of the non-giant
 * observations (1/0), observations laid out, lane-steps laid out (observations + padding), sub-tiles, segments,
 * mean largest number of lanes on one point per wave instruction, mean largest number of lanes on one camera accumulator
 * per wave, observations of giant points (they stay on the point-major path), mean extra LDS cycles per 8-byte read of a
 * point array (bank conflicts), the same per atomic add }. */
int miba_plan_product_stats(const miba_problem* p, double* out10);

#ifdef __cplusplus
}
#endif
#endif

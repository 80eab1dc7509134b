/*
 * ba_oracle.h — CPU fp64 ORACLE for the bundle-adjustment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under bundle_adjustment_amd/, include/ or app/
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it (as the checker / the timed CPU baseline).
 *
 * It restates, in plain C, what the reference reaches through three calls:
 *   BAConstraint::create      (/root/reference/src/BAConstraint.cpp:82-86)
 *   Problem::AddResidualBlock (/root/reference/src/BundleAdjustment.cpp:68-74)
 *   ceres::Solve              (/root/reference/src/BundleAdjustment.cpp:108)
 * The arithmetic behind ceres::Solve lives in ceres-solver, an un-vendored git
 * submodule of the reference (/root/reference/.gitmodules:4-6, libs/ceres-solver is an
 * empty directory, pinned commit unknown, Ceres ~2.0.0 by date).  Its published
 * algorithm (trust_region_minimizer / levenberg_marquardt_strategy /
 * implicit_schur_complement / conjugate_gradients_solver) is restated here from
 * SURVEY.md App. A/B.
 *
 * PARITY PIN: the reference holds no golden vectors or solver tests for this path
 *  and Ceres cannot be built here, so against Ceres itself parity
 * is UNPINNED.  The oracle is pinned instead by fixtures generated in this repo:
 * 50-digit mpmath residual/Jacobian vectors (tests/golden/proj_golden.json,
 * tools/gen_golden.py) and an independent numpy dense-LM restatement
 * (oracle/dense_lm.py, tests/golden/lm_tiny_*.json).
 */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* linear solver selection (mirrors ceres::LinearSolverType use at app/main.cpp:19,25) */
enum {
  BAO_ITERATIVE_SCHUR = 0, /* implicit Schur complement + block-Jacobi PCG (App. B.3) */
  BAO_DENSE_SCHUR = 1,     /* explicit dense Schur complement + Cholesky (exact)      */
  BAO_DENSE_NORMAL = 2     /* full dense normal equations + Cholesky (tiny problems)  */
};

/* ceres::PreconditionerType for ITERATIVE_SCHUR (the reference leaves Ceres' default, JACOBI) */
enum { BAO_JACOBI = 0, BAO_SCHUR_JACOBI = 1 };

typedef struct {
  int max_num_iterations;              /* app/main.cpp:27 sets 80; Ceres default 50 */
  int linear_solver;                   /* BAO_* above */
  int jacobi_scaling;                  /* Ceres default true */
  int max_linear_solver_iterations;    /* 500 */
  int min_linear_solver_iterations;    /* 0 */
  int max_num_consecutive_invalid_steps; /* 5 */
  int num_threads;                     /* OpenMP threads; app/main.cpp:28 sets 4 */
  int preconditioner;                  /* BAO_JACOBI (Ceres default) or BAO_SCHUR_JACOBI, for ITERATIVE_SCHUR */
  double initial_trust_region_radius;  /* 1e4 */
  double max_trust_region_radius;      /* 1e16 */
  double min_trust_region_radius;      /* 1e-32 */
  double min_relative_decrease;        /* 1e-3 */
  double min_lm_diagonal;              /* 1e-6 */
  double max_lm_diagonal;              /* 1e32 */
  double function_tolerance;           /* 1e-6 */
  double gradient_tolerance;           /* 1e-10 */
  double parameter_tolerance;          /* 1e-8 */
  double eta;                          /* 1e-1 */
  /* constant parameter blocks (ceres::Problem::SetParameterBlockConstant, src/BundleAdjustment.cpp:77-79,88-90):
     [n_cam][3] flags for (rotation, translation, intrinsics), non-zero = constant; NULL = all free */
  const unsigned char* cam_const;
  /* robust loss on every residual block (ceres::LossFunction; the reference passes nullptr, src/BundleAdjustment.cpp:69):
     BAO_LOSS_* below with scale a; cost = 1/2 sum rho(|r|^2), residual and Jacobian rows corrected as Ceres' Corrector
     does for rho'' <= 0: both multiplied by sqrt(rho') */
  int loss_type;
  double loss_scale;
  /* camera model: BAO_MODEL_BAL (the 9-parameter BAL camera, default) or BAO_MODEL_REFERENCE, the reference's live functor
     (src/BAConstraint.cpp:15-80: inverse extrinsic, fx fy ox oy, no distortion, observed - projected; bao_project2).  Its
     intrinsics block is constant, as the reference marks it (src/BundleAdjustment.cpp:77): cam_k4 = [n_cam][4] fx fy ox oy;
     the camera vector's entries 6..8 are not used and their columns are zero. */
  int camera_model;
  const double* cam_k4;
} bao_options;
enum { BAO_MODEL_BAL = 0, BAO_MODEL_REFERENCE = 1 };

/* ceres::HuberLoss(a): rho(s) = s (s <= a^2), 2 a sqrt(s) - a^2 otherwise; ceres::CauchyLoss(a): a^2 log(1 + s/a^2) */
enum { BAO_LOSS_NONE = 0, BAO_LOSS_HUBER = 1, BAO_LOSS_CAUCHY = 2 };

typedef struct {
  int iteration;
  int step_is_valid;
  int step_is_successful;
  int linear_solver_iterations;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double gradient_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  double eta;
  double iteration_time_s;
  double model_cost_change;
} bao_iteration;

/* termination codes (mirror ceres::TerminationType) */
enum { BAO_CONVERGENCE = 0, BAO_NO_CONVERGENCE = 1, BAO_FAILURE = 2 };

typedef struct {
  int termination;
  int num_iterations;      /* entries written to iters[] (including iteration 0) */
  int num_successful;
  int num_unsuccessful;
  double initial_cost;
  double final_cost;
  double total_time_s;
} bao_summary;

void bao_default_options(bao_options* o);

/* One observation of the reference's live functor, BAConstraint::operator() (src/BAConstraint.cpp:15-80), restated line by
 * line: R = AngleAxisToRotationMatrix(rot) (:35, Ceres' two branches), extr = [R | tr; 0 0 0 1] (:41-44), P = extr^-1 [X; 1]
 * (:46-57), Pimg = P.xy / P.z (:67-68, no sign flip), the radial factor is computed and not used (:69-70), Pproj = K Pimg
 * (:72), residual = observation - Pproj (:73).  cam6 = rot[3], tr[3]; k4 = fx fy ox oy.
 * Jc18: 2 x 9 row-major with columns rot 0..2, tr 3..5 and three zero columns (the intrinsics block is constant); Jp6: 2 x 3.
 * Derivative of the inverse as autodiff sees it: d(R^-1) = -R^-1 dR R^-1. */
void bao_project2(const double* cam6, const double* k4, const double* X, const double* obs2, double* r2, double* Jc18, double* Jp6);
/* bao_evaluate for either model: k4 NULL = BAL, else the reference's model with cams = [n_cam][9] (entries 6..8 unused) */
double bao_evaluate2(int n_cam, int n_pt, int64_t n_obs, const double* cams, const double* k4, const double* pts,
                     const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                     double* r, double* Jc, double* Jp, double* grad_cam, double* grad_pt);

/* One observation of the 9-parameter BAL camera (SURVEY.md App. A.2-A.4;
 * block order follows BAConstraint.cpp:16,83: point, rot, tr, intrinsics).
 * cam9 = [w0 w1 w2 t0 t1 t2 f k1 k2];  r = pred - obs.
 * Jc (2x9 row-major) and Jp (2x3 row-major) may be NULL. */
void bao_project(const double* cam9, const double* X, const double* obs2,
                 double* r2, double* Jc18, double* Jp6);

/* Whole-problem evaluation in the CALLER'S observation order.
 * Any of r (2*n_obs), Jc (18*n_obs), Jp (6*n_obs), grad_cam (9*n_cam), grad_pt (3*n_pt)
 * may be NULL.  Returns cost = 1/2 sum |r|^2.  Unscaled Jacobian. */
double bao_evaluate(int n_cam, int n_pt, int64_t n_obs, const double* cams,
                    const double* pts, const int32_t* obs_cam, const int32_t* obs_pt,
                    const double* obs_xy, double* r, double* Jc, double* Jp,
                    double* grad_cam, double* grad_pt);

/* Levenberg-Marquardt solve (App. B).  cams/pts are updated in place with the best
 * state.  iters may be NULL; at most max_iters entries are written.  Returns 0 on
 * success, <0 on bad arguments. */
int bao_solve(int n_cam, int n_pt, int64_t n_obs, double* cams, double* pts,
              const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
              const bao_options* opt, bao_iteration* iters, int max_iters,
              bao_summary* summary);

/* One damped linear solve at the given state and radius, for kernel-level parity:
 * returns the (scaled-coordinate) solution y of (J'J + D'D) y = J'r, i.e. BEFORE the
 * LM negation, for cameras (9*n_cam) and points (3*n_pt); also the Schur rhs
 * (9*n_cam), and S*v for a probe vector v (9*n_cam) if v != NULL. */
int bao_linear_step(int n_cam, int n_pt, int64_t n_obs, const double* cams,
                    const double* pts, const int32_t* obs_cam, const int32_t* obs_pt,
                    const double* obs_xy, const bao_options* opt, double radius,
                    const double* scale_cam, const double* scale_pt,
                    double* y_cam, double* y_pt, double* rhs, const double* v,
                    double* Sv, int* n_cg, double* model_cost_change);

/* Jacobi column scale 1/(1+sqrt(sum J^2)) at the given state (App. B.1). */
void bao_jacobi_scale(int n_cam, int n_pt, int64_t n_obs, const double* cams,
                      const double* pts, const int32_t* obs_cam, const int32_t* obs_pt,
                      const double* obs_xy, double* scale_cam, double* scale_pt);

/* The reference loader's initial-state noise (src/BalDataloader.cpp:113-157):
 * Marsaglia polar normals from glibc rand(), draw order R0,R1,R2,T0,T1,T2 per camera
 * then x,y,z per point.  cameras_f: 9 floats per camera, points_f: 3 floats per point.
 * Calls srand(seed) first when seed >= 0 (the reference never seeds => seed 1). */
void bao_bal_initialize(int n_cam, int n_pt, const float* cameras_f, const float* points_f,
                        double sigma_cam, double sigma_x, long seed,
                        double* R, double* T, double* intrinsics, double* X);

#ifdef __cplusplus
}
#endif
#endif

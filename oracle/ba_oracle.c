/*
 * ba_oracle.c — CPU fp64 ORACLE (test infrastructure; see ba_oracle.h header).
 *
 * Restates, in plain C, the algorithm the reference reaches through
 *   /root/reference/src/BAConstraint.cpp:15-86   (residual blocks: point3, rot3, tr3, intr6)
 *   /root/reference/src/BundleAdjustment.cpp:39-113 (problem wiring, ceres::Solve)
 *   /root/reference/src/BalDataloader.cpp:113-157   (initial-state noise)
 * with the camera model fixed by BASELINE.json north_star / SURVEY.md App. A.2 (the
 * 9-parameter BAL camera; the live functor at BAConstraint.cpp:15-80 is the pinhole
 * debugging variant, its BAL form survives commented at :89-128).
 * ceres-solver is an un-vendored submodule of the reference (.gitmodules:4-6, commit
 * unknown, ~2.0.0): its published LM / implicit-Schur / CG algorithm is restated from
 * SURVEY.md App. B.  Against Ceres itself: PARITY UNPINNED (no fixtures in the reference).
 */
#define _POSIX_C_SOURCE 200809L
#include "ba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void bao_default_options(bao_options* o) {
  /* SURVEY.md App. B.4 (Ceres defaults) */
  memset(o, 0, sizeof(*o));
  o->max_num_iterations = 50;
  o->linear_solver = BAO_ITERATIVE_SCHUR;
  o->jacobi_scaling = 1;
  o->max_linear_solver_iterations = 500;
  o->min_linear_solver_iterations = 0;
  o->max_num_consecutive_invalid_steps = 5;
  o->num_threads = 1;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->eta = 1e-1;
  o->cam_const = NULL;
  o->loss_type = BAO_LOSS_NONE;
  o->camera_model = BAO_MODEL_BAL;
  o->cam_k4 = NULL;
  o->loss_scale = 1.0;
}

/* ------------------------------------------------------------------------- */
/* Camera model + analytic Jacobian                */
/* ------------------------------------------------------------------------- */

static inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

void bao_project(const double* cam9, const double* X, const double* obs2, double* r2,
                 double* Jc18, double* Jp6) {
  const double* w = cam9;
  const double* t = cam9 + 3;
  const double f = cam9[6], k1 = cam9[7], k2 = cam9[8];
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double RX[3], R[9] /* row-major */, dRX[3][3] /* dRX[j] = d(RX)/dw_j */;
  const int want_j = (Jc18 != NULL) || (Jp6 != NULL);

  if (theta2 > DBL_EPSILON) {
    /* ceres::AngleAxisRotatePoint main branch (rotation.h), App. A.3 */
    const double theta = sqrt(theta2);
    const double c = cos(theta), s = sin(theta);
    const double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    double a[3];
    cross3(k, X, a);
    const double b = k[0] * X[0] + k[1] * X[1] + k[2] * X[2];
    /* 1 - cos without cancellation at small theta (App. A.4 note); equals Ceres' (1 - costheta)
       to a few ulp of 1-c, which keeps the Jacobian within 1e-12 of the mpmath fixtures */
    const double omc = (c > 0.0) ? s * s / (1.0 + c) : 1.0 - c;
    for (int i = 0; i < 3; ++i) RX[i] = X[i] * c + a[i] * s + k[i] * b * omc;
    if (want_j) {
      /* R = c I + s [k]x + (1-c) k k' */
      R[0] = c + k[0] * k[0] * omc;
      R[1] = -k[2] * s + k[0] * k[1] * omc;
      R[2] = k[1] * s + k[0] * k[2] * omc;
      R[3] = k[2] * s + k[0] * k[1] * omc;
      R[4] = c + k[1] * k[1] * omc;
      R[5] = -k[0] * s + k[1] * k[2] * omc;
      R[6] = -k[1] * s + k[0] * k[2] * omc;
      R[7] = k[0] * s + k[1] * k[2] * omc;
      R[8] = c + k[2] * k[2] * omc;
      /* direct differentiation of Rodrigues (App. A.4) */
      for (int j = 0; j < 3; ++j) {
        double dk[3], dkxX[3];
        for (int i = 0; i < 3; ++i) dk[i] = ((i == j ? 1.0 : 0.0) - k[i] * k[j]) / theta;
        cross3(dk, X, dkxX);
        const double dkX = dk[0] * X[0] + dk[1] * X[1] + dk[2] * X[2];
        const double kj = k[j]; /* d theta / d w_j */
        for (int i = 0; i < 3; ++i) {
          dRX[j][i] = -X[i] * s * kj + a[i] * c * kj + k[i] * b * s * kj + dkxX[i] * s +
                      (dk[i] * b + k[i] * dkX) * omc;
        }
      }
    }
  } else {
    /* near-zero branch: RX = X + w x X; autodiff sees exactly this expression */
    double wxX[3];
    cross3(w, X, wxX);
    for (int i = 0; i < 3; ++i) RX[i] = X[i] + wxX[i];
    if (want_j) {
      R[0] = 1.0; R[1] = -w[2]; R[2] = w[1];
      R[3] = w[2]; R[4] = 1.0; R[5] = -w[0];
      R[6] = -w[1]; R[7] = w[0]; R[8] = 1.0;
      for (int j = 0; j < 3; ++j) {
        double e[3] = {0, 0, 0};
        e[j] = 1.0;
        cross3(e, X, dRX[j]);
      }
    }
  }
  const double P[3] = {RX[0] + t[0], RX[1] + t[1], RX[2] + t[2]};
  const double iz = 1.0 / P[2];
  const double px = -P[0] * iz, py = -P[1] * iz;
  const double n = px * px + py * py;
  const double d = 1.0 + n * (k1 + k2 * n);
  r2[0] = f * d * px - obs2[0];
  r2[1] = f * d * py - obs2[1];
  if (!want_j) return;

  /* dp/dP (2x3) */
  const double dpdP[6] = {-iz, 0.0, P[0] * iz * iz, 0.0, -iz, P[1] * iz * iz};
  /* dpred/dp = f (d I + 2 g p p'), g = k1 + 2 k2 n */
  const double g = k1 + 2.0 * k2 * n;
  const double M[4] = {f * (d + 2.0 * g * px * px), f * (2.0 * g * px * py),
                       f * (2.0 * g * px * py), f * (d + 2.0 * g * py * py)};
  double A[6];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 3; ++j) A[i * 3 + j] = M[i * 2 + 0] * dpdP[j] + M[i * 2 + 1] * dpdP[3 + j];
  if (Jp6) {
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j)
        Jp6[i * 3 + j] = A[i * 3 + 0] * R[0 + j] + A[i * 3 + 1] * R[3 + j] + A[i * 3 + 2] * R[6 + j];
  }
  if (Jc18) {
    for (int i = 0; i < 2; ++i) {
      for (int j = 0; j < 3; ++j) {
        Jc18[i * 9 + j] = A[i * 3 + 0] * dRX[j][0] + A[i * 3 + 1] * dRX[j][1] + A[i * 3 + 2] * dRX[j][2];
        Jc18[i * 9 + 3 + j] = A[i * 3 + j];
      }
    }
    Jc18[0 * 9 + 6] = d * px;          Jc18[1 * 9 + 6] = d * py;
    Jc18[0 * 9 + 7] = f * n * px;      Jc18[1 * 9 + 7] = f * n * py;
    Jc18[0 * 9 + 8] = f * n * n * px;  Jc18[1 * 9 + 8] = f * n * n * py;
  }
}

/* R (row-major) and dR/dw_j of ceres::AngleAxisToRotationMatrix, both branches */
static void rotation_and_derivative(const double* w, double* R, double dR[3][9]) {
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (theta2 > DBL_EPSILON) {
    const double theta = sqrt(theta2);
    const double c = cos(theta), s = sin(theta);
    const double omc = (c > 0.0) ? s * s / (1.0 + c) : 1.0 - c;
    const double k[3] = {w[0] / theta, w[1] / theta, w[2] / theta};
    R[0] = c + k[0] * k[0] * omc;        R[1] = -k[2] * s + k[0] * k[1] * omc; R[2] = k[1] * s + k[0] * k[2] * omc;
    R[3] = k[2] * s + k[0] * k[1] * omc; R[4] = c + k[1] * k[1] * omc;         R[5] = -k[0] * s + k[1] * k[2] * omc;
    R[6] = -k[1] * s + k[0] * k[2] * omc; R[7] = k[0] * s + k[1] * k[2] * omc; R[8] = c + k[2] * k[2] * omc;
    for (int j = 0; j < 3; ++j) {
      double dk[3];
      for (int i = 0; i < 3; ++i) dk[i] = ((i == j ? 1.0 : 0.0) - k[i] * k[j]) / theta;
      const double kj = k[j];   /* d theta / d w_j */
      /* d/dw_j [ c I + s [k]x + (1-c) k k' ] */
      const double v[3] = {c * kj * k[0] + s * dk[0], c * kj * k[1] + s * dk[1], c * kj * k[2] + s * dk[2]};
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q)
          dR[j][3 * r + q] = (r == q ? -s * kj : 0.0) + s * kj * k[r] * k[q] + omc * (dk[r] * k[q] + k[r] * dk[q]);
      dR[j][1] += -v[2]; dR[j][2] += v[1];
      dR[j][3] += v[2];  dR[j][5] += -v[0];
      dR[j][6] += -v[1]; dR[j][7] += v[0];
    }
  } else {
    R[0] = 1.0; R[1] = -w[2]; R[2] = w[1];
    R[3] = w[2]; R[4] = 1.0; R[5] = -w[0];
    R[6] = -w[1]; R[7] = w[0]; R[8] = 1.0;
    memset(dR, 0, sizeof(double) * 27);
    dR[0][5] = -1.0; dR[0][7] = 1.0;
    dR[1][2] = 1.0;  dR[1][6] = -1.0;
    dR[2][1] = -1.0; dR[2][3] = 1.0;
  }
}

static void inv3(const double* a, double* inv) {
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
  inv[0] = c00 * id; inv[1] = (a[2] * a[7] - a[1] * a[8]) * id; inv[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  inv[3] = c01 * id; inv[4] = (a[0] * a[8] - a[2] * a[6]) * id; inv[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  inv[6] = c02 * id; inv[7] = (a[1] * a[6] - a[0] * a[7]) * id; inv[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

void bao_project2(const double* cam6, const double* k4, const double* X, const double* obs2, double* r2, double* Jc18, double* Jp6) {
  const double* t = cam6 + 3;
  const double fx = k4[0], fy = k4[1], ox = k4[2], oy = k4[3];
  double R[9], dR[3][9], Ri[9];
  rotation_and_derivative(cam6, R, dR);           /* src/BAConstraint.cpp:35 */
  inv3(R, Ri);                                    /* :46 (the 4x4 inverse of [R t; 0 1] is [R^-1, -R^-1 t; 0 1]) */
  const double Y[3] = {X[0] - t[0], X[1] - t[1], X[2] - t[2]};
  double P[3];
  for (int i = 0; i < 3; ++i) P[i] = Ri[3 * i] * Y[0] + Ri[3 * i + 1] * Y[1] + Ri[3 * i + 2] * Y[2];   /* :57 */
  const double iz = 1.0 / P[2];
  const double px = P[0] * iz, py = P[1] * iz;    /* :67-68 */
  r2[0] = obs2[0] - (fx * px + ox);               /* :72-76 */
  r2[1] = obs2[1] - (fy * py + oy);
  if (!Jc18 && !Jp6) return;
  /* d r / d P */
  const double A[6] = {-fx * iz, 0.0, fx * P[0] * iz * iz, 0.0, -fy * iz, fy * P[1] * iz * iz};
  if (Jp6)
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 3; ++j) Jp6[3 * i + j] = A[3 * i] * Ri[j] + A[3 * i + 1] * Ri[3 + j] + A[3 * i + 2] * Ri[6 + j];
  if (Jc18) {
    memset(Jc18, 0, sizeof(double) * 18);
    for (int j = 0; j < 3; ++j) {
      /* dP/dw_j = -R^-1 dR_j P */
      double u[3], v[3];
      for (int i = 0; i < 3; ++i) u[i] = dR[j][3 * i] * P[0] + dR[j][3 * i + 1] * P[1] + dR[j][3 * i + 2] * P[2];
      for (int i = 0; i < 3; ++i) v[i] = -(Ri[3 * i] * u[0] + Ri[3 * i + 1] * u[1] + Ri[3 * i + 2] * u[2]);
      for (int i = 0; i < 2; ++i) {
        Jc18[9 * i + j] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
        /* dP/dt_j = -R^-1 e_j */
        Jc18[9 * i + 3 + j] = -(A[3 * i] * Ri[j] + A[3 * i + 1] * Ri[3 + j] + A[3 * i + 2] * Ri[6 + j]);
      }
    }
  }
}

/* model dispatch of the solver: per-camera intrinsics of the reference model ride in the problem (NULL: BAL) */
static const double* g_unused_k4 = NULL;
static inline void project_any(const double* k4, int c, const double* cams, const double* X, const double* obs, double* r, double* F,
                               double* E) {
  if (k4) bao_project2(cams + 9 * (size_t)c, k4 + 4 * (size_t)c, X, obs, r, F, E);
  else bao_project(cams + 9 * (size_t)c, X, obs, r, F, E);
}

double bao_evaluate2(int n_cam, int n_pt, int64_t n_obs, const double* cams, const double* k4, const double* pts,
                     const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                     double* r, double* Jc, double* Jp, double* grad_cam, double* grad_pt) {
  double cost = 0.0;
  (void)g_unused_k4;
  if (grad_cam) memset(grad_cam, 0, sizeof(double) * 9 * (size_t)n_cam);
  if (grad_pt) memset(grad_pt, 0, sizeof(double) * 3 * (size_t)n_pt);
  const int want_j = Jc || Jp || grad_cam || grad_pt;
  for (int64_t i = 0; i < n_obs; ++i) {
    double rr[2], jc[18], jp[6];
    const int c = obs_cam[i], p = obs_pt[i];
    project_any(k4, c, cams, pts + 3 * (size_t)p, obs_xy + 2 * i, rr, want_j ? jc : NULL, want_j ? jp : NULL);
    cost += rr[0] * rr[0] + rr[1] * rr[1];
    if (r) { r[2 * i] = rr[0]; r[2 * i + 1] = rr[1]; }
    if (Jc) memcpy(Jc + 18 * i, jc, sizeof(jc));
    if (Jp) memcpy(Jp + 6 * i, jp, sizeof(jp));
    if (grad_cam)
      for (int k = 0; k < 9; ++k) grad_cam[9 * (size_t)c + k] += jc[k] * rr[0] + jc[9 + k] * rr[1];
    if (grad_pt)
      for (int k = 0; k < 3; ++k) grad_pt[3 * (size_t)p + k] += jp[k] * rr[0] + jp[3 + k] * rr[1];
  }
  return 0.5 * cost;
}

double bao_evaluate(int n_cam, int n_pt, int64_t n_obs, const double* cams, const double* pts,
                    const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                    double* r, double* Jc, double* Jp, double* grad_cam, double* grad_pt) {
  double cost = 0.0;
  if (grad_cam) memset(grad_cam, 0, sizeof(double) * 9 * (size_t)n_cam);
  if (grad_pt) memset(grad_pt, 0, sizeof(double) * 3 * (size_t)n_pt);
  const int want_j = Jc || Jp || grad_cam || grad_pt;
  for (int64_t i = 0; i < n_obs; ++i) {
    double rr[2], jc[18], jp[6];
    const int c = obs_cam[i], p = obs_pt[i];
    bao_project(cams + 9 * (size_t)c, pts + 3 * (size_t)p, obs_xy + 2 * i, rr,
                want_j ? jc : NULL, want_j ? jp : NULL);
    cost += rr[0] * rr[0] + rr[1] * rr[1];
    if (r) { r[2 * i] = rr[0]; r[2 * i + 1] = rr[1]; }
    if (Jc) memcpy(Jc + 18 * i, jc, sizeof(jc));
    if (Jp) memcpy(Jp + 6 * i, jp, sizeof(jp));
    if (grad_cam)
      for (int k = 0; k < 9; ++k) grad_cam[9 * (size_t)c + k] += jc[k] * rr[0] + jc[9 + k] * rr[1];
    if (grad_pt)
      for (int k = 0; k < 3; ++k) grad_pt[3 * (size_t)p + k] += jp[k] * rr[0] + jp[3 + k] * rr[1];
  }
  return 0.5 * cost;
}

/* ------------------------------------------------------------------------- */
/* Problem in point-sorted (CSR) form with stored Jacobian                    */
/* ------------------------------------------------------------------------- */

typedef struct {
  int n_cam, n_pt;
  int64_t n_obs;
  int n_thr;
  int64_t* pt_off; /* n_pt + 1 */
  int32_t* cam;    /* [n_obs] camera of sorted obs */
  double* oxy;     /* [2 n_obs] */
  double *r, *F, *E; /* [2],[18],[6] per sorted obs; F,E column-scaled */
  double* tmp2;    /* [2 n_obs] scratch (Ceres' tmp_rows_) */
  unsigned char *cam_active, *pt_active;
  double* tbuf;    /* per-thread camera accumulators: n_thr * 81 * n_cam */
  int loss_type;   /* BAO_LOSS_*: applied to every residual block */
  double loss_a;
  const double* k4; /* [n_cam][4] fx fy ox oy: the reference's camera model (bao_project2); NULL: BAL */
} prob_t;

/* ceres::HuberLoss / CauchyLoss ::Evaluate on s = |r|^2: rho(s) and rho'(s) (rho'' <= 0 for both, so Ceres'
 * Corrector scales residual and Jacobian rows by sqrt(rho') and nothing else) */
static void loss_eval(int type, double a, double s, double* rho, double* rho1) {
  const double b = a * a;
  if (type == BAO_LOSS_HUBER && s > b) {
    const double r = sqrt(s);
    *rho = 2.0 * a * r - b;
    *rho1 = fmax(DBL_MIN, a / r);
  } else if (type == BAO_LOSS_CAUCHY) {
    const double sum = 1.0 + s / b;
    *rho = b * log(sum);
    *rho1 = fmax(DBL_MIN, 1.0 / sum);
  } else {
    *rho = s;
    *rho1 = 1.0;
  }
}

static int prob_build(prob_t* P, int n_cam, int n_pt, int64_t n_obs, const int32_t* obs_cam,
                      const int32_t* obs_pt, const double* obs_xy, int n_thr) {
  memset(P, 0, sizeof(*P));
  P->n_cam = n_cam; P->n_pt = n_pt; P->n_obs = n_obs; P->n_thr = n_thr < 1 ? 1 : n_thr;
  P->pt_off = (int64_t*)calloc((size_t)n_pt + 1, sizeof(int64_t));
  P->cam = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_obs ? n_obs : 1));
  P->oxy = (double*)malloc(sizeof(double) * 2 * (size_t)(n_obs ? n_obs : 1));
  P->r = (double*)malloc(sizeof(double) * 2 * (size_t)(n_obs ? n_obs : 1));
  P->F = (double*)malloc(sizeof(double) * 18 * (size_t)(n_obs ? n_obs : 1));
  P->E = (double*)malloc(sizeof(double) * 6 * (size_t)(n_obs ? n_obs : 1));
  P->tmp2 = (double*)malloc(sizeof(double) * 2 * (size_t)(n_obs ? n_obs : 1));
  P->cam_active = (unsigned char*)calloc((size_t)n_cam + 1, 1);
  P->pt_active = (unsigned char*)calloc((size_t)n_pt + 1, 1);
  P->tbuf = (double*)malloc(sizeof(double) * (size_t)P->n_thr * 81 * (size_t)(n_cam ? n_cam : 1));
  if (!P->pt_off || !P->cam || !P->oxy || !P->r || !P->F || !P->E || !P->tmp2 || !P->tbuf) return -1;
  for (int64_t i = 0; i < n_obs; ++i) {
    if (obs_pt[i] < 0 || obs_pt[i] >= n_pt || obs_cam[i] < 0 || obs_cam[i] >= n_cam) return -2;
    P->pt_off[obs_pt[i] + 1]++;
  }
  for (int p = 0; p < n_pt; ++p) P->pt_off[p + 1] += P->pt_off[p];
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n_pt + 1));
  memcpy(cur, P->pt_off, sizeof(int64_t) * ((size_t)n_pt + 1));
  for (int64_t i = 0; i < n_obs; ++i) { /* stable: file order within a point */
    const int64_t s = cur[obs_pt[i]]++;
    P->cam[s] = obs_cam[i];
    P->oxy[2 * s] = obs_xy[2 * i];
    P->oxy[2 * s + 1] = obs_xy[2 * i + 1];
    P->cam_active[obs_cam[i]] = 1;
    P->pt_active[obs_pt[i]] = 1;
  }
  free(cur);
  return 0;
}

static void prob_free(prob_t* P) {
  free(P->pt_off); free(P->cam); free(P->oxy); free(P->r); free(P->F); free(P->E);
  free(P->tmp2); free(P->cam_active); free(P->pt_active); free(P->tbuf);
}

static int thr_id(void) {
#ifdef _OPENMP
  return omp_get_thread_num();
#else
  return 0;
#endif
}

/* sum per-thread camera buffers (width doubles per camera) into out (+=) in thread order */
static void reduce_tbuf(const prob_t* P, int width, double* out) {
  const size_t n = (size_t)width * (size_t)P->n_cam;
#pragma omp parallel for schedule(static) num_threads(P->n_thr)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    double s = 0.0;
    for (int t = 0; t < P->n_thr; ++t) s += P->tbuf[(size_t)t * n + (size_t)i];
    out[i] += s;
  }
}

/* residuals + Jacobian at (cams, pts); gradient from the UNSCALED Jacobian, then the
 * stored Jacobian is column-scaled (App. B.1: "J <- J diag(scale)" after every evaluation). */
static double prob_evaluate(prob_t* P, const double* cams, const double* pts, const double* sc,
                            const double* sp, double* grad_cam, double* grad_pt) {
  double cost = 0.0;
  const size_t ncw = 9 * (size_t)P->n_cam;
  memset(P->tbuf, 0, sizeof(double) * (size_t)P->n_thr * ncw);
#pragma omp parallel for schedule(static) reduction(+ : cost) num_threads(P->n_thr)
  for (int p = 0; p < P->n_pt; ++p) {
    double* gc = P->tbuf + (size_t)thr_id() * ncw;
    double gp[3] = {0, 0, 0};
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const int c = P->cam[s];
      double* r = P->r + 2 * s;
      double* F = P->F + 18 * s;
      double* E = P->E + 6 * s;
      project_any(P->k4, c, cams, pts + 3 * (size_t)p, P->oxy + 2 * s, r, F, E);
      if (P->loss_type != BAO_LOSS_NONE) {
        double rho, rho1;
        loss_eval(P->loss_type, P->loss_a, r[0] * r[0] + r[1] * r[1], &rho, &rho1);
        const double w = sqrt(rho1);
        r[0] *= w; r[1] *= w;
        for (int k = 0; k < 18; ++k) F[k] *= w;
        for (int k = 0; k < 6; ++k) E[k] *= w;
        cost += rho;
      } else {
        cost += r[0] * r[0] + r[1] * r[1];
      }
      for (int k = 0; k < 9; ++k) gc[9 * (size_t)c + k] += F[k] * r[0] + F[9 + k] * r[1];
      for (int k = 0; k < 3; ++k) gp[k] += E[k] * r[0] + E[3 + k] * r[1];
      if (sc)
        for (int k = 0; k < 9; ++k) { F[k] *= sc[9 * (size_t)c + k]; F[9 + k] *= sc[9 * (size_t)c + k]; }
      if (sp)
        for (int k = 0; k < 3; ++k) { E[k] *= sp[3 * (size_t)p + k]; E[3 + k] *= sp[3 * (size_t)p + k]; }
    }
    if (grad_pt) for (int k = 0; k < 3; ++k) grad_pt[3 * (size_t)p + k] = gp[k];
  }
  if (grad_cam) {
    memset(grad_cam, 0, sizeof(double) * ncw);
    reduce_tbuf(P, 9, grad_cam);
  }
  return 0.5 * cost;
}

static double prob_cost(const prob_t* P, const double* cams, const double* pts) {
  double cost = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : cost) num_threads(P->n_thr)
  for (int p = 0; p < P->n_pt; ++p) {
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      double r[2];
      project_any(P->k4, P->cam[s], cams, pts + 3 * (size_t)p, P->oxy + 2 * s, r, NULL, NULL);
      double rho = r[0] * r[0] + r[1] * r[1], rho1;
      if (P->loss_type != BAO_LOSS_NONE) loss_eval(P->loss_type, P->loss_a, rho, &rho, &rho1);
      cost += rho;
    }
  }
  return 0.5 * cost;
}

/* squared column norms of the stored Jacobian */
static void prob_colnorm2(prob_t* P, double* nc, double* np) {
  const size_t ncw = 9 * (size_t)P->n_cam;
  memset(P->tbuf, 0, sizeof(double) * (size_t)P->n_thr * ncw);
#pragma omp parallel for schedule(static) num_threads(P->n_thr)
  for (int p = 0; p < P->n_pt; ++p) {
    double* gc = P->tbuf + (size_t)thr_id() * ncw;
    double a[3] = {0, 0, 0};
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const double* F = P->F + 18 * s;
      const double* E = P->E + 6 * s;
      const size_t c = (size_t)P->cam[s];
      for (int k = 0; k < 9; ++k) gc[9 * c + k] += F[k] * F[k] + F[9 + k] * F[9 + k];
      for (int k = 0; k < 3; ++k) a[k] += E[k] * E[k] + E[3 + k] * E[3 + k];
    }
    for (int k = 0; k < 3; ++k) np[3 * (size_t)p + k] = a[k];
  }
  memset(nc, 0, sizeof(double) * ncw);
  reduce_tbuf(P, 9, nc);
}

/* ------------------------------------------------------------------------- */
/* Small dense helpers                                                        */
/* ------------------------------------------------------------------------- */

/* in-place lower Cholesky of n x n row-major A (reads lower triangle); 0 ok, -1 not SPD */
static int chol_lower(int n, double* A, int lda) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * lda + j];
    for (int k = 0; k < j; ++k) d -= A[j * lda + k] * A[j * lda + k];
    if (!(d > 0.0) || !isfinite(d)) return -1;
    d = sqrt(d);
    A[j * lda + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * lda + j];
      for (int k = 0; k < j; ++k) s -= A[i * lda + k] * A[j * lda + k];
      A[i * lda + j] = s / d;
    }
  }
  return 0;
}

static void chol_solve(int n, const double* L, int lda, double* b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * lda + k] * b[k];
    b[i] = s / L[i * lda + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * lda + i] * b[k];
    b[i] = s / L[i * lda + i];
  }
}

/* Ainv = A^-1 through Cholesky (Eigen llt().solve(I) in Ceres); n <= 9 */
static int spd_inverse(int n, const double* A, double* Ainv) {
  double L[81], col[9];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) L[i * n + j] = A[i * n + j];
  if (chol_lower(n, L, n)) return -1;
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < n; ++i) col[i] = (i == j) ? 1.0 : 0.0;
    chol_solve(n, L, n, col);
    for (int i = 0; i < n; ++i) Ainv[i * n + j] = col[i];
  }
  return 0;
}

/* large dense Cholesky (lower, row-major), blocked just enough for N ~ 3000 */
static int chol_lower_big(int64_t n, double* A, int n_thr) {
  for (int64_t j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int64_t k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !isfinite(d)) return -1;
    d = sqrt(d);
    A[j * n + j] = d;
    const double* Aj = A + j * n;
#pragma omp parallel for schedule(static) num_threads(n_thr) if (n - j > 256)
    for (int64_t i = j + 1; i < n; ++i) {
      double* Ai = A + i * n;
      double s = Ai[j];
      for (int64_t k = 0; k < j; ++k) s -= Ai[k] * Aj[k];
      Ai[j] = s / d;
    }
  }
  return 0;
}

static void chol_solve_big(int64_t n, const double* L, double* b) {
  for (int64_t i = 0; i < n; ++i) {
    double s = b[i];
    for (int64_t k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int64_t k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

/* ------------------------------------------------------------------------- */
/* Schur machinery (App. B.3)                                                 */
/* ------------------------------------------------------------------------- */

typedef struct {
  double* Cinv;  /* [9 n_pt]  (E'E + Dp^2)^-1 */
  double* gp;    /* [3 n_pt]  E'b */
  double* Minv;  /* [81 n_cam] (F'F + Dc^2)^-1 block diagonal (JACOBI preconditioner) */
  double* FtF;   /* [81 n_cam] F'F + Dc^2 */
  double* rhs;   /* [9 n_cam] */
  double* Dc2;   /* [9 n_cam] */
  double* Dp2;   /* [3 n_pt] */
} schur_t;

static int schur_alloc(schur_t* S, int n_cam, int n_pt) {
  S->Cinv = (double*)malloc(sizeof(double) * 9 * ((size_t)n_pt + 1));
  S->gp = (double*)malloc(sizeof(double) * 3 * ((size_t)n_pt + 1));
  S->Minv = (double*)malloc(sizeof(double) * 81 * ((size_t)n_cam + 1));
  S->FtF = (double*)malloc(sizeof(double) * 81 * ((size_t)n_cam + 1));
  S->rhs = (double*)malloc(sizeof(double) * 9 * ((size_t)n_cam + 1));
  S->Dc2 = (double*)malloc(sizeof(double) * 9 * ((size_t)n_cam + 1));
  S->Dp2 = (double*)malloc(sizeof(double) * 3 * ((size_t)n_pt + 1));
  return (S->Cinv && S->gp && S->Minv && S->FtF && S->rhs && S->Dc2 && S->Dp2) ? 0 : -1;
}
static void schur_free(schur_t* S) {
  free(S->Cinv); free(S->gp); free(S->Minv); free(S->FtF); free(S->rhs); free(S->Dc2); free(S->Dp2);
}

/* ImplicitSchurComplement::Init: block inverses + rhs = F'(b - E C^-1 E'b). */
static int schur_init(prob_t* P, schur_t* S, int schur_jacobi) {
  const int n_cam = P->n_cam, n_pt = P->n_pt;
  const size_t w81 = 81 * (size_t)n_cam;
  int fail = 0;
  /* pass 1: per point C^-1, g_p; per camera F'F */
  memset(P->tbuf, 0, sizeof(double) * (size_t)P->n_thr * w81);
#pragma omp parallel for schedule(static) num_threads(P->n_thr) reduction(| : fail)
  for (int p = 0; p < n_pt; ++p) {
    double* acc = P->tbuf + (size_t)thr_id() * w81;
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const double* E = P->E + 6 * s;
      const double* F = P->F + 18 * s;
      const double* r = P->r + 2 * s;
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) C[i * 3 + j] += E[i] * E[j] + E[3 + i] * E[3 + j];
        g[i] += E[i] * r[0] + E[3 + i] * r[1];
      }
      double* B = acc + 81 * (size_t)P->cam[s];
      for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) B[i * 9 + j] += F[i] * F[j] + F[9 + i] * F[9 + j];
    }
    for (int i = 0; i < 3; ++i) C[i * 3 + i] += S->Dp2[3 * (size_t)p + i];
    if (spd_inverse(3, C, S->Cinv + 9 * (size_t)p)) fail |= 1;
    for (int i = 0; i < 3; ++i) S->gp[3 * (size_t)p + i] = g[i];
  }
  if (fail) return -1;
  memset(S->FtF, 0, sizeof(double) * w81);
  reduce_tbuf(P, 81, S->FtF);
  /* SCHUR_JACOBI (ceres::PreconditionerType, SchurJacobiPreconditioner): the preconditioner is the block diagonal
     of S itself, B_c - sum_{obs of c} (E'F)' C_p^-1 (E'F), instead of the block diagonal B_c of the camera part */
  double* Sd = NULL;
  if (schur_jacobi) {
    Sd = (double*)calloc(w81 ? w81 : 1, sizeof(double));
    if (!Sd) return -1;
    memset(P->tbuf, 0, sizeof(double) * (size_t)P->n_thr * w81);
#pragma omp parallel for schedule(static) num_threads(P->n_thr)
    for (int p = 0; p < n_pt; ++p) {
      double* acc = P->tbuf + (size_t)thr_id() * w81;
      const double* Ci = S->Cinv + 9 * (size_t)p;
      for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
        const double* E = P->E + 6 * s;
        const double* F = P->F + 18 * s;
        double G[27], T[27];
        for (int a = 0; a < 3; ++a)
          for (int k = 0; k < 9; ++k) G[9 * a + k] = E[a] * F[k] + E[3 + a] * F[9 + k];
        for (int a = 0; a < 3; ++a)
          for (int k = 0; k < 9; ++k) T[9 * a + k] = Ci[a * 3] * G[k] + Ci[a * 3 + 1] * G[9 + k] + Ci[a * 3 + 2] * G[18 + k];
        double* D = acc + 81 * (size_t)P->cam[s];
        for (int i = 0; i < 9; ++i)
          for (int j = 0; j < 9; ++j) D[i * 9 + j] += G[i] * T[j] + G[9 + i] * T[9 + j] + G[18 + i] * T[18 + j];
      }
    }
    reduce_tbuf(P, 81, Sd);
  }
#pragma omp parallel for schedule(static) num_threads(P->n_thr) reduction(| : fail)
  for (int c = 0; c < n_cam; ++c) {
    double* B = S->FtF + 81 * (size_t)c;
    for (int i = 0; i < 9; ++i) B[i * 9 + i] += S->Dc2[9 * (size_t)c + i];
    if (Sd) {
      double M[81];
      for (int i = 0; i < 81; ++i) M[i] = B[i] - Sd[81 * (size_t)c + i];
      if (spd_inverse(9, M, S->Minv + 81 * (size_t)c)) fail |= 1;
    } else if (spd_inverse(9, B, S->Minv + 81 * (size_t)c)) fail |= 1;
  }
  free(Sd);
  if (fail) return -1;
  /* pass 2: rhs = F'(b - E C^-1 g_p) */
  const size_t w9 = 9 * (size_t)n_cam;
  memset(P->tbuf, 0, sizeof(double) * (size_t)P->n_thr * w9);
#pragma omp parallel for schedule(static) num_threads(P->n_thr)
  for (int p = 0; p < n_pt; ++p) {
    double* acc = P->tbuf + (size_t)thr_id() * w9;
    const double* Ci = S->Cinv + 9 * (size_t)p;
    const double* g = S->gp + 3 * (size_t)p;
    double y[3];
    for (int i = 0; i < 3; ++i) y[i] = Ci[i * 3] * g[0] + Ci[i * 3 + 1] * g[1] + Ci[i * 3 + 2] * g[2];
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const double* E = P->E + 6 * s;
      const double* F = P->F + 18 * s;
      const double* r = P->r + 2 * s;
      const double v0 = r[0] - (E[0] * y[0] + E[1] * y[1] + E[2] * y[2]);
      const double v1 = r[1] - (E[3] * y[0] + E[4] * y[1] + E[5] * y[2]);
      double* a = acc + 9 * (size_t)P->cam[s];
      for (int k = 0; k < 9; ++k) a[k] += F[k] * v0 + F[9 + k] * v1;
    }
  }
  memset(S->rhs, 0, sizeof(double) * w9);
  reduce_tbuf(P, 9, S->rhs);
  return 0;
}

/* ImplicitSchurComplement::RightMultiply: y = S x */
static void schur_matvec(prob_t* P, const schur_t* S, const double* x, double* y) {
  const size_t w9 = 9 * (size_t)P->n_cam;
  memset(P->tbuf, 0, sizeof(double) * (size_t)P->n_thr * w9);
#pragma omp parallel for schedule(static) num_threads(P->n_thr)
  for (int p = 0; p < P->n_pt; ++p) {
    double* acc = P->tbuf + (size_t)thr_id() * w9;
    double y2[3] = {0, 0, 0};
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const double* F = P->F + 18 * s;
      const double* E = P->E + 6 * s;
      const double* xc = x + 9 * (size_t)P->cam[s];
      double u0 = 0, u1 = 0;
      for (int k = 0; k < 9; ++k) { u0 += F[k] * xc[k]; u1 += F[9 + k] * xc[k]; }
      P->tmp2[2 * s] = u0; P->tmp2[2 * s + 1] = u1;
      for (int k = 0; k < 3; ++k) y2[k] += E[k] * u0 + E[3 + k] * u1;
    }
    const double* Ci = S->Cinv + 9 * (size_t)p;
    double y3[3];
    for (int i = 0; i < 3; ++i) y3[i] = -(Ci[i * 3] * y2[0] + Ci[i * 3 + 1] * y2[1] + Ci[i * 3 + 2] * y2[2]);
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const double* F = P->F + 18 * s;
      const double* E = P->E + 6 * s;
      const double w0 = P->tmp2[2 * s] + E[0] * y3[0] + E[1] * y3[1] + E[2] * y3[2];
      const double w1 = P->tmp2[2 * s + 1] + E[3] * y3[0] + E[4] * y3[1] + E[5] * y3[2];
      double* a = acc + 9 * (size_t)P->cam[s];
      for (int k = 0; k < 9; ++k) a[k] += F[k] * w0 + F[9 + k] * w1;
    }
  }
  for (size_t i = 0; i < w9; ++i) y[i] = S->Dc2[i] * x[i];
  reduce_tbuf(P, 9, y);
}

/* BackSubstitute: y_p = C^-1 E'(b - F x_c) */
static void schur_backsub(const prob_t* P, const schur_t* S, const double* xc, double* yp) {
#pragma omp parallel for schedule(static) num_threads(P->n_thr)
  for (int p = 0; p < P->n_pt; ++p) {
    double g[3] = {0, 0, 0};
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const double* F = P->F + 18 * s;
      const double* E = P->E + 6 * s;
      const double* r = P->r + 2 * s;
      const double* x = xc + 9 * (size_t)P->cam[s];
      double u0 = 0, u1 = 0;
      for (int k = 0; k < 9; ++k) { u0 += F[k] * x[k]; u1 += F[9 + k] * x[k]; }
      const double v0 = r[0] - u0, v1 = r[1] - u1;
      for (int k = 0; k < 3; ++k) g[k] += E[k] * v0 + E[3 + k] * v1;
    }
    const double* Ci = S->Cinv + 9 * (size_t)p;
    for (int i = 0; i < 3; ++i) yp[3 * (size_t)p + i] = Ci[i * 3] * g[0] + Ci[i * 3 + 1] * g[1] + Ci[i * 3 + 2] * g[2];
  }
}

static double dot_n(size_t n, const double* a, const double* b) {
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

static int is_zero_or_inf(double x) { return x == 0.0 || isinf(x); }

/* ConjugateGradientsSolver::Solve with x0 = 0, JACOBI (block-diagonal F'F) preconditioner.
 * Returns 0 success / 1 no convergence (solution still used) / 2 failure. */
static int schur_pcg(prob_t* P, const schur_t* S, double q_tol, int min_it, int max_it, double* x,
                     int* n_it) {
  const size_t n = 9 * (size_t)P->n_cam;
  const double* b = S->rhs;
  double* r = (double*)malloc(sizeof(double) * n);
  double* p = (double*)malloc(sizeof(double) * n);
  double* z = (double*)malloc(sizeof(double) * n);
  double* tmp = (double*)malloc(sizeof(double) * n);
  int status = 1;
  memset(x, 0, sizeof(double) * n);
  *n_it = 0;
  const double norm_b = sqrt(dot_n(n, b, b));
  if (norm_b == 0.0) { status = 0; goto done; }
  memcpy(r, b, sizeof(double) * n); /* r = b - S*0 */
  double rho = 1.0;
  double Q0 = 0.0; /* -x.(b + r) with x = 0 */
  for (int it = 1;; ++it) {
    *n_it = it;
    /* z = M^-1 r */
    for (int c = 0; c < P->n_cam; ++c) {
      const double* Mi = S->Minv + 81 * (size_t)c;
      const double* rc = r + 9 * (size_t)c;
      for (int i = 0; i < 9; ++i) {
        double s = 0;
        for (int j = 0; j < 9; ++j) s += Mi[i * 9 + j] * rc[j];
        z[9 * (size_t)c + i] = s;
      }
    }
    const double last_rho = rho;
    rho = dot_n(n, r, z);
    if (is_zero_or_inf(rho)) { status = 2; break; }
    if (it == 1) {
      memcpy(p, z, sizeof(double) * n);
    } else {
      const double beta = rho / last_rho;
      if (is_zero_or_inf(beta)) { status = 2; break; }
      for (size_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    double* q = z;
    schur_matvec(P, S, p, q);
    const double pq = dot_n(n, p, q);
    if (pq <= 0.0 || isinf(pq)) { status = 1; break; }
    const double alpha = rho / pq;
    if (isinf(alpha)) { status = 2; break; }
    for (size_t i = 0; i < n; ++i) x[i] += alpha * p[i];
    if (it % 10 == 0) { /* residual_reset_period */
      schur_matvec(P, S, x, tmp);
      for (size_t i = 0; i < n; ++i) r[i] = b[i] - tmp[i];
    } else {
      for (size_t i = 0; i < n; ++i) r[i] -= alpha * q[i];
    }
    double Q1 = 0.0;
    for (size_t i = 0; i < n; ++i) Q1 += x[i] * (b[i] + r[i]);
    Q1 = -Q1;
    const double zeta = it * (Q1 - Q0) / Q1;
    if (zeta < q_tol && it >= min_it) { status = 0; break; }
    Q0 = Q1;
    if (it >= max_it) break;
  }
done:
  free(r); free(p); free(z); free(tmp);
  return status;
}

/* explicit dense Schur complement + Cholesky (stand-in for SPARSE_SCHUR/DENSE_SCHUR) */
static int schur_dense_solve(prob_t* P, const schur_t* S, double* x) {
  const int64_t N = 9 * (int64_t)P->n_cam;
  double* A = (double*)calloc((size_t)(N * N), sizeof(double));
  if (!A) return 2;
  for (int c = 0; c < P->n_cam; ++c)
    for (int i = 0; i < 9; ++i)
      for (int j = 0; j < 9; ++j) A[(9 * (int64_t)c + i) * N + 9 * c + j] = S->FtF[81 * (size_t)c + i * 9 + j];
  for (int p = 0; p < P->n_pt; ++p) {
    const int64_t s0 = P->pt_off[p], s1 = P->pt_off[p + 1];
    const double* Ci = S->Cinv + 9 * (size_t)p;
    for (int64_t a = s0; a < s1; ++a) {
      /* Ga = F_a' E_a (9x3); Ha = Ga C^-1 (9x3) */
      double Ga[27], Ha[27];
      const double* Fa = P->F + 18 * a;
      const double* Ea = P->E + 6 * a;
      for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 3; ++j) Ga[i * 3 + j] = Fa[i] * Ea[j] + Fa[9 + i] * Ea[3 + j];
      for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 3; ++j)
          Ha[i * 3 + j] = Ga[i * 3] * Ci[j] + Ga[i * 3 + 1] * Ci[3 + j] + Ga[i * 3 + 2] * Ci[6 + j];
      for (int64_t b = s0; b < s1; ++b) {
        const double* Fb = P->F + 18 * b;
        const double* Eb = P->E + 6 * b;
        double* blk = A + (9 * (int64_t)P->cam[a]) * N + 9 * (int64_t)P->cam[b];
        for (int i = 0; i < 9; ++i)
          for (int j = 0; j < 9; ++j) {
            double gb0 = Fb[j] * Eb[0] + Fb[9 + j] * Eb[3];
            double gb1 = Fb[j] * Eb[1] + Fb[9 + j] * Eb[4];
            double gb2 = Fb[j] * Eb[2] + Fb[9 + j] * Eb[5];
            blk[i * N + j] -= Ha[i * 3] * gb0 + Ha[i * 3 + 1] * gb1 + Ha[i * 3 + 2] * gb2;
          }
      }
    }
  }
  int rc = chol_lower_big(N, A, P->n_thr);
  if (rc == 0) {
    memcpy(x, S->rhs, sizeof(double) * (size_t)N);
    chol_solve_big(N, A, x);
  }
  free(A);
  return rc ? 2 : 0;
}

/* full dense normal equations (points first, cameras second), tiny problems only */
static int normal_dense_solve(prob_t* P, const schur_t* S, double* xc, double* xp) {
  const int64_t np3 = 3 * (int64_t)P->n_pt, N = np3 + 9 * (int64_t)P->n_cam;
  double* A = (double*)calloc((size_t)(N * N), sizeof(double));
  double* b = (double*)calloc((size_t)N, sizeof(double));
  if (!A || !b) { free(A); free(b); return 2; }
  for (int p = 0; p < P->n_pt; ++p) {
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      double J[2][12];
      int64_t idx[12];
      for (int k = 0; k < 3; ++k) { idx[k] = 3 * (int64_t)p + k; J[0][k] = P->E[6 * s + k]; J[1][k] = P->E[6 * s + 3 + k]; }
      for (int k = 0; k < 9; ++k) { idx[3 + k] = np3 + 9 * (int64_t)P->cam[s] + k; J[0][3 + k] = P->F[18 * s + k]; J[1][3 + k] = P->F[18 * s + 9 + k]; }
      for (int i = 0; i < 12; ++i) {
        b[idx[i]] += J[0][i] * P->r[2 * s] + J[1][i] * P->r[2 * s + 1];
        for (int j = 0; j < 12; ++j) A[idx[i] * N + idx[j]] += J[0][i] * J[0][j] + J[1][i] * J[1][j];
      }
    }
  }
  for (int64_t i = 0; i < np3; ++i) A[i * N + i] += S->Dp2[i];
  for (int64_t i = 0; i < 9 * (int64_t)P->n_cam; ++i) A[(np3 + i) * N + np3 + i] += S->Dc2[i];
  int rc = chol_lower_big(N, A, P->n_thr);
  if (rc == 0) {
    chol_solve_big(N, A, b);
    memcpy(xp, b, sizeof(double) * (size_t)np3);
    memcpy(xc, b + np3, sizeof(double) * 9 * (size_t)P->n_cam);
  }
  free(A); free(b);
  return rc ? 2 : 0;
}

/* LevenbergMarquardtStrategy::ComputeStep without the final negation.
 * Returns 0 ok / 2 linear solver failure. */
static int lm_linear_solve(prob_t* P, schur_t* S, const bao_options* o, const double* diag_c,
                           const double* diag_p, double radius, double* yc, double* yp, int* n_cg) {
  const size_t nc9 = 9 * (size_t)P->n_cam, np3 = 3 * (size_t)P->n_pt;
  for (size_t i = 0; i < nc9; ++i) S->Dc2[i] = diag_c[i] / radius; /* D = sqrt(diag/radius) => D^2 */
  for (size_t i = 0; i < np3; ++i) S->Dp2[i] = diag_p[i] / radius;
  *n_cg = 0;
  if (schur_init(P, S, o->preconditioner == BAO_SCHUR_JACOBI)) return 2;
  int st;
  if (o->linear_solver == BAO_DENSE_NORMAL) {
    st = normal_dense_solve(P, S, yc, yp);
    if (st) return 2;
  } else {
    if (o->linear_solver == BAO_DENSE_SCHUR) {
      st = schur_dense_solve(P, S, yc);
    } else {
      st = schur_pcg(P, S, o->eta, o->min_linear_solver_iterations, o->max_linear_solver_iterations, yc, n_cg);
    }
    if (st == 2) return 2;
    schur_backsub(P, S, yc, yp);
  }
  for (size_t i = 0; i < nc9; ++i) if (!isfinite(yc[i])) return 2;
  for (size_t i = 0; i < np3; ++i) if (!isfinite(yp[i])) return 2;
  return 0;
}

/* model_cost_change = -(J step).(r + J step / 2) with step = -y  (App. B.1) */
static double model_cost_change(const prob_t* P, const double* yc, const double* yp) {
  double acc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : acc) num_threads(P->n_thr)
  for (int p = 0; p < P->n_pt; ++p) {
    const double* y3 = yp + 3 * (size_t)p;
    for (int64_t s = P->pt_off[p]; s < P->pt_off[p + 1]; ++s) {
      const double* F = P->F + 18 * s;
      const double* E = P->E + 6 * s;
      const double* x = yc + 9 * (size_t)P->cam[s];
      double m0 = E[0] * y3[0] + E[1] * y3[1] + E[2] * y3[2];
      double m1 = E[3] * y3[0] + E[4] * y3[1] + E[5] * y3[2];
      for (int k = 0; k < 9; ++k) { m0 += F[k] * x[k]; m1 += F[9 + k] * x[k]; }
      m0 = -m0; m1 = -m1; /* J * step */
      acc += m0 * (P->r[2 * s] + 0.5 * m0) + m1 * (P->r[2 * s + 1] + 0.5 * m1);
    }
  }
  return -acc;
}

static void clamp_diag(const prob_t* P, const bao_options* o, double* dc, double* dp) {
  for (size_t i = 0; i < 9 * (size_t)P->n_cam; ++i) dc[i] = fmin(fmax(dc[i], o->min_lm_diagonal), o->max_lm_diagonal);
  for (size_t i = 0; i < 3 * (size_t)P->n_pt; ++i) dp[i] = fmin(fmax(dp[i], o->min_lm_diagonal), o->max_lm_diagonal);
}

/* constant blocks: zero the entries of a camera-sized vector that belong to constant parameter blocks */
static void mask_const(const prob_t* P, const unsigned char* cam_const, double* vc) {
  if (!cam_const) return;
  for (int c = 0; c < P->n_cam; ++c)
    for (int k = 0; k < 9; ++k)
      if (cam_const[3 * (size_t)c + k / 3]) vc[9 * (size_t)c + k] = 0.0;
}
/* |x|^2 over the free entries of the active blocks */
static double free_sq(const prob_t* P, const unsigned char* cam_const, const double* vc, const double* vp) {
  double s = 0.0;
  for (int c = 0; c < P->n_cam; ++c)
    if (P->cam_active[c])
      for (int k = 0; k < 9; ++k)
        if (!cam_const || !cam_const[3 * (size_t)c + k / 3]) s += vc[9 * (size_t)c + k] * vc[9 * (size_t)c + k];
  for (int p = 0; p < P->n_pt; ++p)
    if (P->pt_active[p]) for (int k = 0; k < 3; ++k) s += vp[3 * (size_t)p + k] * vp[3 * (size_t)p + k];
  return s;
}

/* norms over ACTIVE parameter blocks only (inactive ones are not in the ceres::Problem) */
static double active_sq(const prob_t* P, const double* vc, const double* vp) {
  double s = 0.0;
  for (int c = 0; c < P->n_cam; ++c)
    if (P->cam_active[c]) for (int k = 0; k < 9; ++k) s += vc[9 * (size_t)c + k] * vc[9 * (size_t)c + k];
  for (int p = 0; p < P->n_pt; ++p)
    if (P->pt_active[p]) for (int k = 0; k < 3; ++k) s += vp[3 * (size_t)p + k] * vp[3 * (size_t)p + k];
  return s;
}
static double active_max(const prob_t* P, const double* vc, const double* vp) {
  double m = 0.0;
  for (int c = 0; c < P->n_cam; ++c)
    if (P->cam_active[c]) for (int k = 0; k < 9; ++k) m = fmax(m, fabs(vc[9 * (size_t)c + k]));
  for (int p = 0; p < P->n_pt; ++p)
    if (P->pt_active[p]) for (int k = 0; k < 3; ++k) m = fmax(m, fabs(vp[3 * (size_t)p + k]));
  return m;
}

void bao_jacobi_scale(int n_cam, int n_pt, int64_t n_obs, const double* cams, const double* pts,
                      const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                      double* scale_cam, double* scale_pt) {
  prob_t P;
  if (prob_build(&P, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_xy, 1)) return;
  prob_evaluate(&P, cams, pts, NULL, NULL, NULL, NULL);
  prob_colnorm2(&P, scale_cam, scale_pt);
  for (size_t i = 0; i < 9 * (size_t)n_cam; ++i) scale_cam[i] = 1.0 / (1.0 + sqrt(scale_cam[i]));
  for (size_t i = 0; i < 3 * (size_t)n_pt; ++i) scale_pt[i] = 1.0 / (1.0 + sqrt(scale_pt[i]));
  prob_free(&P);
}

int bao_linear_step(int n_cam, int n_pt, int64_t n_obs, const double* cams, const double* pts,
                    const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
                    const bao_options* opt, double radius, const double* scale_cam,
                    const double* scale_pt, double* y_cam, double* y_pt, double* rhs,
                    const double* v, double* Sv, int* n_cg, double* mcc) {
  prob_t P;
  schur_t S;
  int rc = prob_build(&P, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_xy, opt->num_threads);
  if (rc) return rc;
  P.loss_type = opt->loss_type; P.loss_a = opt->loss_scale;
  if (opt->camera_model == BAO_MODEL_REFERENCE) {
    /* the reference keeps the intrinsics block of every camera constant (src/BundleAdjustment.cpp:77) */
    if (!opt->cam_k4 || !opt->cam_const) { prob_free(&P); return -3; }
    for (int c = 0; c < n_cam; ++c) if (!opt->cam_const[3 * (size_t)c + 2]) { prob_free(&P); return -3; }
    P.k4 = opt->cam_k4;
  }
  if (schur_alloc(&S, n_cam, n_pt)) { prob_free(&P); return -1; }
  double* dc = (double*)malloc(sizeof(double) * 9 * ((size_t)n_cam + 1));
  double* dp = (double*)malloc(sizeof(double) * 3 * ((size_t)n_pt + 1));
  prob_evaluate(&P, cams, pts, scale_cam, scale_pt, NULL, NULL);
  prob_colnorm2(&P, dc, dp);
  clamp_diag(&P, opt, dc, dp);
  int ncg = 0;
  rc = lm_linear_solve(&P, &S, opt, dc, dp, radius, y_cam, y_pt, &ncg);
  if (n_cg) *n_cg = ncg;
  if (rhs) memcpy(rhs, S.rhs, sizeof(double) * 9 * (size_t)n_cam);
  if (v && Sv) schur_matvec(&P, &S, v, Sv);
  if (mcc) *mcc = rc ? 0.0 : model_cost_change(&P, y_cam, y_pt);
  free(dc); free(dp);
  schur_free(&S);
  prob_free(&P);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* Trust-region LM loop (App. B.1, B.2)                                       */
/* ------------------------------------------------------------------------- */

int bao_solve(int n_cam, int n_pt, int64_t n_obs, double* cams, double* pts,
              const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_xy,
              const bao_options* opt, bao_iteration* iters, int max_iters, bao_summary* summary) {
  if (!cams || !pts || !opt || n_cam <= 0 || n_pt <= 0 || n_obs < 0) return -1;
  const double t_start = now_s();
  prob_t P;
  schur_t S;
  int rc = prob_build(&P, n_cam, n_pt, n_obs, obs_cam, obs_pt, obs_xy, opt->num_threads);
  if (rc) return rc;
  P.loss_type = opt->loss_type; P.loss_a = opt->loss_scale;
  if (opt->camera_model == BAO_MODEL_REFERENCE) {
    /* the reference keeps the intrinsics block of every camera constant (src/BundleAdjustment.cpp:77) */
    if (!opt->cam_k4 || !opt->cam_const) { prob_free(&P); return -3; }
    for (int c = 0; c < n_cam; ++c) if (!opt->cam_const[3 * (size_t)c + 2]) { prob_free(&P); return -3; }
    P.k4 = opt->cam_k4;
  }
  if (schur_alloc(&S, n_cam, n_pt)) { prob_free(&P); return -1; }
  const size_t nc9 = 9 * (size_t)n_cam, np3 = 3 * (size_t)n_pt;
  double* xc = (double*)malloc(sizeof(double) * nc9);   /* current x */
  double* xp = (double*)malloc(sizeof(double) * np3);
  double* cc = (double*)malloc(sizeof(double) * nc9);   /* candidate */
  double* cp = (double*)malloc(sizeof(double) * np3);
  double* sc = (double*)malloc(sizeof(double) * nc9);   /* jacobi scale */
  double* sp = (double*)malloc(sizeof(double) * np3);
  double* gc = (double*)malloc(sizeof(double) * nc9);   /* gradient */
  double* gp = (double*)malloc(sizeof(double) * np3);
  double* dc = (double*)malloc(sizeof(double) * nc9);   /* LM diagonal (clamped col norms) */
  double* dp = (double*)malloc(sizeof(double) * np3);
  double* yc = (double*)malloc(sizeof(double) * nc9);   /* linear solution */
  double* yp = (double*)malloc(sizeof(double) * np3);
  memcpy(xc, cams, sizeof(double) * nc9);
  memcpy(xp, pts, sizeof(double) * np3);

  int n_rec = 0, n_succ = 0, n_unsucc = 0;
  int termination = BAO_NO_CONVERGENCE;
  bao_iteration it;
  memset(&it, 0, sizeof(it));

  /* LM strategy state (B.2) */
  double radius = opt->initial_trust_region_radius;
  double decrease_factor = 2.0;
  int reuse_diagonal = 0;
  int consecutive_invalid = 0;

  /* IterationZero */
  double x_norm = sqrt(free_sq(&P, opt->cam_const, xc, xp));
  double x_cost;
  {
    x_cost = prob_evaluate(&P, xc, xp, NULL, NULL, gc, gp);
    mask_const(&P, opt->cam_const, gc);
    if (opt->jacobi_scaling) {
      prob_colnorm2(&P, sc, sp);
      for (size_t i = 0; i < nc9; ++i) sc[i] = 1.0 / (1.0 + sqrt(sc[i]));
      for (size_t i = 0; i < np3; ++i) sp[i] = 1.0 / (1.0 + sqrt(sp[i]));
    } else {
      for (size_t i = 0; i < nc9; ++i) sc[i] = 1.0;
      for (size_t i = 0; i < np3; ++i) sp[i] = 1.0;
    }
    mask_const(&P, opt->cam_const, sc);   /* a zero scale removes the column from the scaled Jacobian */
    if (opt->jacobi_scaling || opt->cam_const) {
      /* scale the stored Jacobian in place */
#pragma omp parallel for schedule(static) num_threads(P.n_thr)
      for (int p = 0; p < n_pt; ++p)
        for (int64_t s = P.pt_off[p]; s < P.pt_off[p + 1]; ++s) {
          const size_t c = (size_t)P.cam[s];
          for (int k = 0; k < 9; ++k) { P.F[18 * s + k] *= sc[9 * c + k]; P.F[18 * s + 9 + k] *= sc[9 * c + k]; }
          for (int k = 0; k < 3; ++k) { P.E[6 * s + k] *= sp[3 * (size_t)p + k]; P.E[6 * s + 3 + k] *= sp[3 * (size_t)p + k]; }
        }
    }
  }
  double minimum_cost = x_cost;
  const double initial_cost = x_cost;
  it.iteration = 0;
  it.cost = x_cost;
  it.gradient_max_norm = active_max(&P, gc, gp);
  it.gradient_norm = sqrt(active_sq(&P, gc, gp));
  it.trust_region_radius = radius;
  it.eta = opt->eta;
  it.iteration_time_s = now_s() - t_start;

  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (it.step_is_successful) {
      ++n_succ;
      if (x_cost < minimum_cost) {
        minimum_cost = x_cost;
        memcpy(cams, xc, sizeof(double) * nc9);
        memcpy(pts, xp, sizeof(double) * np3);
      }
    } else if (it.iteration > 0) {
      ++n_unsucc;
    }
    it.trust_region_radius = radius;
    if (iters && n_rec < max_iters) iters[n_rec] = it;
    ++n_rec;
    if (it.iteration >= opt->max_num_iterations) { termination = BAO_NO_CONVERGENCE; break; }
    if (it.gradient_max_norm <= opt->gradient_tolerance) { termination = BAO_CONVERGENCE; break; }
    if (radius < opt->min_trust_region_radius) { termination = BAO_CONVERGENCE; break; }

    const double t_it = now_s();
    const int iteration = it.iteration + 1;
    const double prev_gmax = it.gradient_max_norm, prev_gnorm = it.gradient_norm;
    memset(&it, 0, sizeof(it));
    it.iteration = iteration;
    it.eta = opt->eta;

    /* ComputeTrustRegionStep */
    if (!reuse_diagonal) {
      prob_colnorm2(&P, dc, dp);
      clamp_diag(&P, opt, dc, dp);
    }
    int n_cg = 0;
    const int ls = lm_linear_solve(&P, &S, opt, dc, dp, radius, yc, yp, &n_cg);
    reuse_diagonal = 1;
    it.linear_solver_iterations = n_cg;
    double mcc = 0.0;
    if (ls == 0) {
      mcc = model_cost_change(&P, yc, yp);
      it.step_is_valid = (mcc > 0.0);
    }
    it.model_cost_change = mcc;
    if (!it.step_is_valid) {
      /* HandleInvalidStep */
      if (++consecutive_invalid >= opt->max_num_consecutive_invalid_steps) {
        termination = BAO_FAILURE;   /* TrustRegionMinimizer returns here: the unfinished iteration is not recorded */
        break;
      }
      radius *= 0.5; /* StepIsInvalid */
      reuse_diagonal = 1;
      it.cost = x_cost; it.cost_change = 0.0;
      it.gradient_max_norm = prev_gmax; it.gradient_norm = prev_gnorm;
      it.step_norm = 0.0; it.relative_decrease = 0.0;
      it.iteration_time_s = now_s() - t_it;
      continue;
    }
    consecutive_invalid = 0;
    /* delta = step .* scale, step = -y ; candidate = x + delta */
    for (size_t i = 0; i < nc9; ++i) cc[i] = xc[i] - yc[i] * sc[i];
    for (size_t i = 0; i < np3; ++i) cp[i] = xp[i] - yp[i] * sp[i];
    double cand_cost = prob_cost(&P, cc, cp);
    if (!isfinite(cand_cost)) cand_cost = DBL_MAX;

    /* ParameterToleranceReached */
    {
      double s2 = 0.0;
      for (int c = 0; c < n_cam; ++c)
        if (P.cam_active[c]) for (int k = 0; k < 9; ++k) { const double d = xc[9 * (size_t)c + k] - cc[9 * (size_t)c + k]; s2 += d * d; }
      for (int p = 0; p < n_pt; ++p)
        if (P.pt_active[p]) for (int k = 0; k < 3; ++k) { const double d = xp[3 * (size_t)p + k] - cp[3 * (size_t)p + k]; s2 += d * d; }
      it.step_norm = sqrt(s2);
    }
    it.cost_change = x_cost - cand_cost;
    it.relative_decrease = it.cost_change / mcc;
    if (it.step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
      termination = BAO_CONVERGENCE;   /* returns before FinalizeIteration...: not recorded, not counted */
      break;
    }
    /* FunctionToleranceReached */
    if (fabs(it.cost_change) <= opt->function_tolerance * x_cost) {
      termination = BAO_CONVERGENCE;   /* returns before FinalizeIteration...: not recorded, not counted */
      break;
    }
    if (it.relative_decrease > opt->min_relative_decrease) {
      /* HandleSuccessfulStep */
      memcpy(xc, cc, sizeof(double) * nc9);
      memcpy(xp, cp, sizeof(double) * np3);
      x_norm = sqrt(free_sq(&P, opt->cam_const, xc, xp));
      x_cost = prob_evaluate(&P, xc, xp, sc, sp, gc, gp);
      mask_const(&P, opt->cam_const, gc);
      it.cost = x_cost;
      it.gradient_max_norm = active_max(&P, gc, gp);
      it.gradient_norm = sqrt(active_sq(&P, gc, gp));
      it.step_is_successful = 1;
      /* StepAccepted */
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * it.relative_decrease - 1.0, 3));
      radius = fmin(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = 0;
    } else {
      it.step_is_successful = 0;
      it.cost = cand_cost;
      it.gradient_max_norm = prev_gmax; it.gradient_norm = prev_gnorm;
      /* StepRejected */
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = 1;
    }
    it.iteration_time_s = now_s() - t_it;
  }

  if (summary) {
    summary->termination = termination;
    summary->num_iterations = n_rec;
    summary->num_successful = n_succ;
    summary->num_unsuccessful = n_unsucc;
    summary->initial_cost = initial_cost;
    summary->final_cost = minimum_cost;
    summary->total_time_s = now_s() - t_start;
  }
  free(xc); free(xp); free(cc); free(cp); free(sc); free(sp); free(gc); free(gp);
  free(dc); free(dp); free(yc); free(yp);
  schur_free(&S);
  prob_free(&P);
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Reference loader's initial noise (src/BalDataloader.cpp:113-157)           */
/* ------------------------------------------------------------------------- */

static double rand_double(void) { return (double)rand() / RAND_MAX; }

static double rand_normal(void) {
  double x1, x2, w;
  do {
    x1 = 2.0 * rand_double() - 1.0;
    x2 = 2.0 * rand_double() - 1.0;
    w = x1 * x1 + x2 * x2;
  } while (w >= 1.0 || w == 0.0);
  w = sqrt((-2.0 * log(w)) / w);
  return x1 * w;
}

void bao_bal_initialize(int n_cam, int n_pt, const float* cameras_f, const float* points_f,
                        double sigma_cam, double sigma_x, long seed, double* R, double* T,
                        double* intrinsics, double* X) {
  if (seed >= 0) srand((unsigned)seed);
  for (int i = 0; i < n_cam; ++i) {
    const float* cam = cameras_f + 9 * (size_t)i;
    for (int k = 0; k < 3; ++k) R[3 * i + k] = cam[k] + rand_normal() * sigma_cam;
    for (int k = 0; k < 3; ++k) T[3 * i + k] = cam[3 + k] + rand_normal() * sigma_cam;
    intrinsics[6 * i + 0] = cam[6];
    intrinsics[6 * i + 1] = cam[6];
    intrinsics[6 * i + 2] = 0;
    intrinsics[6 * i + 3] = 0;
    intrinsics[6 * i + 4] = cam[7];
    intrinsics[6 * i + 5] = cam[8];
  }
  for (int i = 0; i < n_pt; ++i)
    for (int k = 0; k < 3; ++k) X[3 * i + k] = points_f[3 * (size_t)i + k] + rand_normal() * sigma_x;
}

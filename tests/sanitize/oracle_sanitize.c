/* The CPU oracle under AddressSanitizer + UBSan: every linear solver, both preconditioners, constant
 * blocks and both loss functions on a small random problem.  Built and run by tests/test_sanitizers.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "ba_oracle.h"

static double urand(void) { return (double)rand() / RAND_MAX; }

int main(void) {
  enum { NC = 6, NP = 40, PER = 4 };
  const int n_obs = NP * PER;
  double* cams = malloc(sizeof(double) * 9 * NC);
  double* pts = malloc(sizeof(double) * 3 * NP);
  int32_t* oc = malloc(sizeof(int32_t) * n_obs);
  int32_t* op = malloc(sizeof(int32_t) * n_obs);
  double* oxy = malloc(sizeof(double) * 2 * n_obs);
  srand(7);
  for (int c = 0; c < NC; ++c) {
    double* k = cams + 9 * c;
    for (int j = 0; j < 3; ++j) k[j] = 0.1 * (urand() - 0.5);
    k[3] = urand() - 0.5; k[4] = urand() - 0.5; k[5] = -6.0 - urand();
    k[6] = 800.0; k[7] = -0.05; k[8] = 0.01;
  }
  for (int p = 0; p < NP; ++p)
    for (int j = 0; j < 3; ++j) pts[3 * p + j] = 2.0 * (urand() - 0.5);
  for (int p = 0, i = 0; p < NP; ++p)
    for (int k = 0; k < PER; ++k, ++i) {
      oc[i] = (p + k) % NC; op[i] = p;
      double r[2], zero[2] = {0, 0};
      bao_project(cams + 9 * oc[i], pts + 3 * p, zero, r, NULL, NULL);
      oxy[2 * i] = r[0] + urand() - 0.5 + (i % 9 == 0 ? 30.0 : 0.0);
      oxy[2 * i + 1] = r[1] + urand() - 0.5;
    }
  double* c0 = malloc(sizeof(double) * 9 * NC);
  double* p0 = malloc(sizeof(double) * 3 * NP);
  unsigned char cc[3 * NC] = {0};
  cc[0] = cc[1] = cc[2] = 1; cc[3 * 2 + 2] = 1;
  int bad = 0;
  for (int solver = 0; solver < 3; ++solver)
    for (int variant = 0; variant < 5; ++variant) {
      for (int i = 0; i < 9 * NC; ++i) c0[i] = cams[i] + (i % 9 < 6 ? 0.002 * (urand() - 0.5) : 0.0);
      for (int i = 0; i < 3 * NP; ++i) p0[i] = pts[i] + 0.02 * (urand() - 0.5);
      bao_options o;
      bao_default_options(&o);
      o.max_num_iterations = 6; o.linear_solver = solver; o.num_threads = 2;
      if (variant == 1) o.preconditioner = BAO_SCHUR_JACOBI;
      if (variant == 2) o.cam_const = cc;
      if (variant == 3) { o.loss_type = BAO_LOSS_HUBER; o.loss_scale = 1.0; }
      if (variant == 4) { o.loss_type = BAO_LOSS_CAUCHY; o.loss_scale = 2.0; }
      bao_iteration its[16];
      bao_summary s;
      const int rc = bao_solve(NC, NP, n_obs, c0, p0, oc, op, oxy, &o, its, 16, &s);
      if (rc != 0 || !(s.final_cost <= s.initial_cost) || !isfinite(s.final_cost)) {
        fprintf(stderr, "solver %d variant %d: rc %d cost %g -> %g\n", solver, variant, rc, s.initial_cost, s.final_cost);
        ++bad;
      }
    }
  double sc[9 * NC], sp[3 * NP];
  bao_jacobi_scale(NC, NP, n_obs, cams, pts, oc, op, oxy, sc, sp);
  free(cams); free(pts); free(oc); free(op); free(oxy); free(c0); free(p0);
  if (bad) return 1;
  printf("oracle sanitize ok\n");
  return 0;
}

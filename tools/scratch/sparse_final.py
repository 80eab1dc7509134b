import time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bundle_adjustment_amd import miba, synth
def rel(a, b):
    m = np.abs(b).max(); return np.abs(a - b).max() / m
n_cam, n_pt, n_obs, seed = synth.CONFIGS["final-13682"]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
cams0, pts0 = synth.perturb(pr, 1e-3, 0.05, seed=seed + 1)
s = miba.Solver(device_id=0)
s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
for radius in (1e4, 1e2):
    ex = s.linear_step(radius, miba.default_options(linear_solver_type=miba.SPARSE_SCHUR))
    t0 = time.time(); ex = s.linear_step(radius, miba.default_options(linear_solver_type=miba.SPARSE_SCHUR)); t1 = time.time()
    r = s.linear_step(radius, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, max_linear_solver_iterations=1), probe=ex["y_cam"])
    print("radius", radius, "exact: %.3f s, residual |S y - rhs| / |rhs| = %.3e" % (t1 - t0, np.abs(r["Sv"] - ex["rhs"]).max() / np.abs(ex["rhs"]).max()), flush=True)
    for eta in (1e-12, 1e-16, 1e-20):
        t0 = time.time()
        it = s.linear_step(radius, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, eta=eta, max_linear_solver_iterations=5000))
        t1 = time.time()
        r = s.linear_step(radius, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, max_linear_solver_iterations=1), probe=it["y_cam"])
        print("  pcg eta %g: %d it, %.3f s, diff to exact %.3e, residual %.3e, mcc rel diff %.3e" % (eta, it["n_cg"], t1 - t0, rel(it["y_cam"], ex["y_cam"]),
              np.abs(r["Sv"] - it["rhs"]).max() / np.abs(it["rhs"]).max(), abs(it["model_cost_change"] - ex["model_cost_change"]) / abs(ex["model_cost_change"])), flush=True)
kt = s.kernel_times() if hasattr(s, "kernel_times") else None
print(kt)
s.close()

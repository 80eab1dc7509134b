"""Matrix-free kernels against the stored-Jacobian kernels at full size: right-hand side, S v, PCG count, LM costs."""
import os, sys, json, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
cams0, pts0 = synth.perturb(pr, 1e-3, 0.05, seed=seed + 1)
probe = np.random.default_rng(1).normal(size=(n_cam, 9))
res = {}
for mf in ("1", "0"):
    os.environ["MIBA_MF"] = mf
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    out = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR), probe=probe)
    its, summ = s.solve(miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, max_num_iterations=8,
                                             function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
    s.close()
    res[mf] = (out, its, summ)
a, b = res["1"], res["0"]
rel = lambda x, y: float(np.abs(x - y).max() / np.abs(y).max())
print(json.dumps({"config": cfg, "rhs_rel": rel(a[0]["rhs"], b[0]["rhs"]), "Sv_rel": rel(a[0]["Sv"], b[0]["Sv"]),
                  "n_cg": [a[0]["n_cg"], b[0]["n_cg"]], "y_cam_rel": rel(a[0]["y_cam"], b[0]["y_cam"]),
                  "mcc_rel": abs(a[0]["model_cost_change"] - b[0]["model_cost_change"]) / abs(b[0]["model_cost_change"]),
                  "cost_rel_per_iteration": [abs(x["cost"] - y["cost"]) / y["cost"] for x, y in zip(a[1], b[1])],
                  "pcg_per_iteration": [[x["linear_solver_iterations"] for x in a[1]], [y["linear_solver_iterations"] for y in b[1]]],
                  "solve_s": [a[2]["total_time_in_seconds"], b[2]["total_time_in_seconds"]]}))

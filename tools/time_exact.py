"""Exact (dense Schur + Cholesky) vs iterative (implicit-Schur PCG) LM on the small synthetic configs (GPU).
usage: time_exact.py [config ...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth

cfgs = sys.argv[1:] or ["ladybug-49", "trafalgar-257", "dubrovnik-356"]
for cfg in cfgs:
    n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
    pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
    cams0, pts0 = synth.perturb(pr, 1e-3, 0.05, seed=seed + 1)
    row = {"config": cfg, "n_cam": n_cam, "n_obs": int(pr["obs_cam"].size)}
    for name, lst in (("iterative", miba.ITERATIVE_SCHUR), ("exact", miba.DENSE_SCHUR)):
        s = miba.Solver(device_id=0)
        s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
        kw = {"pcg_check_every": int(os.environ["CHECK_EVERY"])} if "CHECK_EVERY" in os.environ else {}
        opt = miba.default_options(linear_solver_type=lst, max_num_iterations=20, function_tolerance=0.0,
                                   parameter_tolerance=0.0, gradient_tolerance=0.0, **kw)
        s.solve(miba.default_options(linear_solver_type=lst, max_num_iterations=2))   # warm-up (allocations, first launches)
        s.set_state(cams0, pts0)
        t0 = time.perf_counter()
        its, summ = s.solve(opt)
        dt = time.perf_counter() - t0
        s.close()
        row[name] = {"lm_iterations": len(its) - 1, "ms_per_lm_iteration": 1e3 * dt / max(1, len(its) - 1),
                     "linear_iterations": summ["num_linear_solver_iterations"], "final_cost": summ["final_cost"],
                     "linear_solver_s": summ["linear_solver_time_in_seconds"]}
    print(json.dumps(row))

#!/usr/bin/env python3
"""Numbers for BASELINE.md section 2: per configuration, LM iterations/s of the HIP path (1 GPU) and of the CPU baseline
(oracle/ba_oracle.c, OpenMP, the host share of a one-GPU job) on the same synthetic stand-in, same initial state
(ground truth + observation noise), same options (LM, ITERATIVE_SCHUR, JACOBI, eta 0.1, tolerances off), 3 repetitions,
median.  usage (GPU box): python tools/baseline_table.py > profiles/r2_baseline_table.json"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bundle_adjustment_amd import miba, synth
from oracle import oracle as O

cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
rows = {}
for name, n_it, cpu_it in (("ladybug-49", 20, 20), ("trafalgar-257", 20, 20), ("dubrovnik-356", 20, 10), ("final-13682", 20, 4),
                           ("synthetic-50k", 6, 0)):
    n_cam, n_pt, n_obs, seed = synth.CONFIGS[name]
    pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
    cams0, pts0 = synth.perturb(pr, 0.0, 0.0, seed=seed + 1)
    kw = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    s = miba.Solver(device_id=0)
    t0 = time.time()
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    t_set = time.time() - t0
    rec = {"n_cam": n_cam, "n_pt": n_pt, "n_obs": n_obs, "set_problem_s": round(t_set, 3)}
    for label, lst in (("iterative", miba.ITERATIVE_SCHUR), ("sparse_schur", miba.SPARSE_SCHUR)):
        vals = []
        for rep in range(3):
            s.set_state(cams0, pts0)
            its, summ = s.solve(miba.default_options(max_num_iterations=n_it, linear_solver_type=lst, pcg_check_every=4, **kw))
            n = len(its) - 1
            vals.append(n / summ["total_time_in_seconds"])
        rec["hip_" + label] = {"lm_it_per_s": float(np.median(vals)), "values": vals, "lm_iterations": n,
                               "pcg_iterations": summ["num_linear_solver_iterations"], "final_cost": summ["final_cost"],
                               "exact_linear_solver": summ["exact_linear_solver"]}
    s.close()
    if cpu_it:
        vals = []
        for rep in range(3):
            _, _, oits, osumm = O.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                                        O.default_options(max_num_iterations=cpu_it, linear_solver=O.ITERATIVE_SCHUR, num_threads=cores, **kw))
            t = sum(r["iteration_time_s"] for r in oits[1:])
            vals.append((len(oits) - 1) / t)
        rec["cpu_iterative"] = {"lm_it_per_s": float(np.median(vals)), "values": vals, "lm_iterations": len(oits) - 1, "threads": cores,
                                "final_cost": osumm["final_cost"]}
    rows[name] = rec
    print(name, json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(rows, indent=1))

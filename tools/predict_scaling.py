"""Measured per-GPU components of a point-sharded run, taken on ONE GPU (GPU box): for N = 1, 2, 4, 8 the sub-problem that rank
N // 2 of an N-way plan owns (its points and their observations, all cameras: exactly the tiles that rank would hold) is solved
by a one-rank handle with a fixed number of PCG iterations per LM iteration; the kernel times of that run are what one GPU of
the N-GPU job spends outside communication.  DESIGN.md section 6 turns them into the predicted 1/2/4/8 curve.
usage: predict_scaling.py [config] [pcg iterations per LM iteration]  -> one JSON line per N"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
n_pcg = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
cams0, pts0 = synth.perturb(pr, 0.0, 0.0, seed=seed + 1)
oc, op, oxy = pr["obs_cam"], pr["obs_pt"], pr["obs_xy"]
for N in (1, 2, 4, 8):
    if N == 1:
        sub_pts, sub_oc, sub_op, sub_oxy = pts0, oc, op, oxy
    else:
        owner = miba.plan_point_owner(n_cam, n_pt, oc, op, oxy, N)
        mine = owner == N // 2
        new_id = np.full(n_pt, -1, dtype=np.int32)
        new_id[mine] = np.arange(int(mine.sum()), dtype=np.int32)
        keep = mine[op]
        sub_pts, sub_oc, sub_op, sub_oxy = pts0[mine], oc[keep], new_id[op[keep]], oxy[keep]
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, sub_pts, sub_oc, sub_op, sub_oxy)
    opt = miba.default_options(max_num_iterations=6, linear_solver_type=miba.ITERATIVE_SCHUR, function_tolerance=0.0, gradient_tolerance=0.0,
                               parameter_tolerance=0.0, min_linear_solver_iterations=n_pcg, max_linear_solver_iterations=n_pcg)
    s.solve(opt)                       # warm-up
    s.set_state(cams0, sub_pts)
    s.reset_kernel_times()
    its, summ = s.solve(opt)
    kt = s.kernel_times()
    s.close()
    avg = lambda k: kt["total_ms"][k] / max(1, kt["launches"][k])
    n_it = len(its) - 1
    t_it = sum(r["iteration_time_in_seconds"] for r in its[1:]) / n_it
    n_cg = sum(r["linear_solver_iterations"] for r in its[1:]) / n_it
    per_lm = avg(miba.K_SCHUR_INIT) + avg(miba.K_SCHUR_INIT_CAM) + avg(miba.K_BACKSUB) + avg(miba.K_EVAL_COST)
    print(json.dumps({"config": cfg, "n_gpus_modelled": N, "shard": N // 2, "observations": int(len(sub_oc)), "points": int(len(sub_pts)),
                      "tiles": kt["n_tiles"], "pcg_iterations_per_lm_iteration": n_cg, "ms_per_lm_iteration": 1e3 * t_it,
                      "product_ms": avg(miba.K_SCHUR_MATVEC), "tail_ms": avg(miba.K_CAM_REDUCE),
                      "K2a_ms": avg(miba.K_SCHUR_INIT), "K2b_ms": avg(miba.K_SCHUR_INIT_CAM), "K4_ms": avg(miba.K_BACKSUB), "K1p_ms": avg(miba.K_EVAL_COST),
                      "once_per_lm_iteration_kernels_ms": per_lm,
                      "rest_ms_per_lm_iteration": 1e3 * t_it - n_cg * (avg(miba.K_SCHUR_MATVEC) + avg(miba.K_CAM_REDUCE)) - per_lm}), flush=True)

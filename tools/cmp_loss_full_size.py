import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from bundle_adjustment_amd import miba, synth
from oracle import oracle as O
n_cam, n_pt, n_obs, seed = synth.CONFIGS["final-13682"]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
cams0, pts0 = synth.perturb(pr, 1e-3, 0.05, seed=seed + 1)
kw = dict(max_num_iterations=5, eta=0.1, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
for loss in ((0, 1.0), (1, 1.0), (2, 2.0)):
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    its, summ = s.solve(miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, loss_type=loss[0], loss_scale=loss[1], **kw))
    s.close()
    _, _, oits, osumm = O.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                                O.default_options(linear_solver=O.ITERATIVE_SCHUR, loss_type=loss[0], loss_scale=loss[1], num_threads=16, **kw))
    print("loss", loss, flush=True)
    for a, b in zip(its, oits):
        print(a["iteration"], "%.12e %.12e %.2e" % (a["cost"], b["cost"], abs(a["cost"] - b["cost"]) / b["cost"]), a["linear_solver_iterations"], b["linear_solver_iterations"],
              a["step_is_successful"], b["step_is_successful"], flush=True)

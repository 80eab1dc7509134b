import time, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bundle_adjustment_amd import miba, synth
n_cam, n_pt, n_obs, seed = synth.CONFIGS["final-13682"]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
cams0, pts0 = synth.perturb(pr, 1e-3, 0.05, seed=seed + 1)
s = miba.Solver(device_id=0)
s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
for i in range(4):
    t0 = time.time(); ex = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.SPARSE_SCHUR)); t1 = time.time()
    print("step %.3f s" % (t1 - t0), flush=True)
s.close()

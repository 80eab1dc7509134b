import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth
pr = synth.make_problem(40, 3000, 14000, seed=11, shuffle_obs=True)
cams0, pts0 = synth.perturb(pr, 0.01, 0.1, seed=12)
rng = np.random.default_rng(5)
probe = rng.normal(size=(40, 9))
res = {}
for mf in ("0", "1", "1"):
    os.environ["MIBA_MF"] = mf
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    out = s.linear_step(1e4, miba.default_options(max_linear_solver_iterations=3), probe=probe)
    s.close()
    res.setdefault(mf, []).append(out["Sv"])
a = res["0"][0]
for b in res["1"]:
    e = np.abs(b - a) / np.abs(a).max()
    print("max rel err", e.max(), "rot", e[:, :3].max(), "tr", e[:, 3:6].max(), "intr", e[:, 6:].max())
    print("bad cams", np.flatnonzero(e.max(1) > 1e-9)[:20], "of 40")
print("mf run-to-run", np.abs(res["1"][0] - res["1"][1]).max())

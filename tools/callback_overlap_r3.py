"""What a per-iteration state pull costs the solver (update_state_every_iteration + callbacks, src/BundleAdjustment.cpp:104-105):
LM solve of a named configuration with (a) no callback, (b) a callback between iterations that pulls the state with
miba_get_state (what the round-2 mirror did), (c) the same callback beside the solver (options.async_callbacks = 1).
usage: callback_overlap_r3.py [config] [iterations]"""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
cams0, pts0 = synth.perturb(pr, 0.0, 0.0, seed=seed + 1)
s = miba.Solver(device_id=0)
s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
R, T, K = miba.split_cams(cams0)
R, T, K, X = [np.ascontiguousarray(a, dtype=np.float64).copy() for a in (R, T, K, pts0)]
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
pulls = []


def pull(it):
    t = time.time()
    rc = miba.lib().miba_get_state(s._h, dp(R), dp(T), dp(K), dp(X))
    pulls.append(time.time() - t)
    return 0 if rc == 0 else 1


out = {}
for tag, cb, a in (("warm_up", None, 0), ("no_callback", None, 0), ("callback_between_iterations", pull, 0), ("callback_beside_the_solver", pull, 1)):
    s.set_state(cams0, pts0)
    opt = miba.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    opt.async_callbacks = a
    pulls.clear()
    t = time.time()
    its, summ = s.solve(opt, callback=cb)
    wall = time.time() - t
    out[tag] = dict(wall_s=round(wall, 4), solver_s=round(summ["total_time_in_seconds"], 4), iterations=len(its) - 1,
                    final_cost=summ["final_cost"], mean_pull_ms=round(1e3 * float(np.mean(pulls)), 2) if pulls else None)
s.close()
out["async_over_no_callback"] = round(out["callback_beside_the_solver"]["wall_s"] / out["no_callback"]["wall_s"], 3)
out["sync_over_no_callback"] = round(out["callback_between_iterations"]["wall_s"] / out["no_callback"]["wall_s"], 3)
print(json.dumps(out))

"""Time the implicit-Schur product in isolation on a synthetic config (GPU).  usage: time_k3.py [config] [reps]"""
import os, sys, json, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
cams0, pts0 = synth.perturb(pr, 1e-3, 0.05, seed=seed + 1)
s = miba.Solver(device_id=0)
s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
probe = np.random.default_rng(1).normal(size=(n_cam, 9))
# (PCG_ITS iterations of the PCG: 12 take in a residual-reset iteration, so every instance of pcg_tail runs under the counters)
out = s.linear_step(1e4, miba.default_options(max_linear_solver_iterations=int(os.environ.get("PCG_ITS", "2")), eta=1e-30), probe=probe)
ms = [s.time_matvec(probe, reps) for _ in range(3)]
kt = s.kernel_times()
print(json.dumps({"tile_chunks": os.environ.get("MIBA_TILE_CHUNKS", "4"), "matvec_ms": ms,
                  "Sv_norm": float(np.linalg.norm(out["Sv"])), "slots": kt["n_slots"], "tiles": kt["n_tiles"]}))

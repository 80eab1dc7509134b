"""Time-to-cost-tolerance as SURVEY 8(d) defines it: C* = final cost of the CPU baseline (the oracle, same options and
initial state, Ceres' default tolerances, at most 80 iterations); reported is the HIP solver time at the first LM
iteration whose cost is <= (1 + 1e-6) C*, next to the oracle's own time to the same cost.
usage: ttt_vs_oracle.py [config] [threads] [sigma_cam sigma_x]   (default start: the noise-free one of bench.py's headline)
bench.py reads the record this prints from profiles/r3_cstar_<config>.json as the committed C* of SURVEY 8(d)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth
from oracle import oracle as O

cfg = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
sc = float(sys.argv[3]) if len(sys.argv) > 4 else 0.0
sx = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
cams0, pts0 = synth.perturb(pr, sc, sx, seed=seed + 1)
t0 = time.time()
_, _, oits, osumm = O.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                            O.default_options(max_num_iterations=80, linear_solver=O.ITERATIVE_SCHUR, num_threads=thr))
t_cpu = time.time() - t0
cstar = osumm["final_cost"]
print("oracle: %d iterations, C* = %.9e, %.1f s wall" % (len(oits) - 1, cstar, t_cpu), file=sys.stderr, flush=True)
s = miba.Solver(device_id=0)
s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
s.solve(miba.default_options(max_num_iterations=2, linear_solver_type=miba.ITERATIVE_SCHUR))   # warm-up
s.set_state(cams0, pts0)
its, summ = s.solve(miba.default_options(max_num_iterations=80, linear_solver_type=miba.ITERATIVE_SCHUR))
s.close()
hit = next((r for r in its if r["cost"] <= (1.0 + 1e-6) * cstar), None)
ocum = np.cumsum([r["iteration_time_s"] for r in oits])
ohit = next((i for i, r in enumerate(oits) if r["cost"] <= (1.0 + 1e-6) * cstar), None)
print(json.dumps({"config": cfg, "sigma_cam": sc, "sigma_x": sx, "seed": seed, "eta": 0.1,
                  "provenance": "tools/ttt_vs_oracle.py on the GPU box: oracle/ba_oracle.c (ITERATIVE_SCHUR, JACOBI, Ceres' default tolerances, at most "
                                "80 iterations) from the same initial state as bench.py's headline",
                  "cpu_termination_code": osumm.get("termination"),
                  "c_star_cpu_baseline": cstar, "cpu_threads": thr, "cpu_iterations": len(oits) - 1,
                  "cpu_seconds_to_tolerance": float(ocum[ohit]) if ohit is not None else None, "cpu_total_seconds": t_cpu,
                  "hip_seconds_to_tolerance": hit["cumulative_time_in_seconds"] if hit else None,
                  "hip_iteration": hit["iteration"] if hit else None, "hip_final_cost": summ["final_cost"],
                  "hip_iterations": summ["num_iterations"], "hip_total_seconds": summ["total_time_in_seconds"],
                  "final_cost_relative_difference": abs(summ["final_cost"] - cstar) / cstar}))

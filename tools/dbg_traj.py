import sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import miba, synth
from oracle import oracle as O
pr = synth.make_problem(40, 3000, 14000, seed=11, shuffle_obs=True)
cams0, pts0 = synth.perturb(pr, 0.01, 0.1, seed=12)
for name, lst, olst, eta in (("exact", miba.SPARSE_SCHUR, O.DENSE_SCHUR, 0.1), ("iter", miba.ITERATIVE_SCHUR, O.ITERATIVE_SCHUR, 0.1), ("iter-tight", miba.ITERATIVE_SCHUR, O.ITERATIVE_SCHUR, 1e-12)):
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    its, summ = s.solve(miba.default_options(max_num_iterations=12, linear_solver_type=lst, eta=eta, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
    s.close()
    _, _, oits, osumm = O.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"], O.default_options(max_num_iterations=12, linear_solver=olst, eta=eta, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0))
    print(name, summ["message"], summ["termination_type"], osumm["termination"])
    for i in range(max(len(its), len(oits))):
        a = its[i] if i < len(its) else None
        b = oits[i] if i < len(oits) else None
        f = lambda d: "%2d %.10e g=%.3e st=%.3e rho=%.4f rad=%.3e cg=%3d ok=%d/%d" % (d["iteration"], d["cost"], d["gradient_max_norm"], d["step_norm"], d["relative_decrease"], d["trust_region_radius"], d["linear_solver_iterations"], d["step_is_valid"], d["step_is_successful"]) if d else "-"
        print("  H", f(a)); print("  O", f(b))

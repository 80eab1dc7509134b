"""Worker of tests/test_gpu_multirank.py: one rank of a multi-rank world on GPU 0, all-reduces through the
shared-memory TEST transport (MIBA_TEST_SHM_COMM).  usage: _multirank_worker.py RANK WORLD OUT.npz medium|giant"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bundle_adjustment_amd import miba, synth  # noqa: E402


def problem(kind):
    if kind == "medium":
        pr = synth.make_problem(40, 3000, 14000, seed=11, shuffle_obs=True)
        cams0, pts0 = synth.perturb(pr, 0.01, 0.1, seed=12)
        return cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"]
    # "giant": ordinary points plus one point seen by more cameras than a wavelet holds (giant-tile kernels)
    rng = np.random.default_rng(3)
    pr = synth.make_problem(150, 400, 2500, seed=31)
    cams = pr["cams"]
    pts = np.concatenate([pr["pts"], np.array([[0.2, -0.1, 0.3]])])
    cs = rng.choice(150, size=140, replace=False).astype(np.int32)
    ps = np.full(140, 400, dtype=np.int32)
    xy, z = synth.project(cams, pts, cs, ps)
    ok = z < 0
    oc = np.concatenate([pr["obs_cam"], cs[ok]]); op = np.concatenate([pr["obs_pt"], ps[ok]])
    oxy = np.concatenate([pr["obs_xy"], xy[ok] + rng.normal(0, 0.5, xy[ok].shape)])
    cams0 = cams.copy(); cams0[:, 0:6] += rng.normal(0, 0.003, (150, 6))
    pts0 = pts + rng.normal(0, 0.05, pts.shape)
    return cams0, pts0, oc, op, oxy


def run(rank, world, kind):
    cams0, pts0, oc, op, oxy = problem(kind)
    s = miba.Solver(device_id=0, rank=rank, world_size=world)
    s.set_problem(cams0, pts0, oc, op, oxy)
    out = {}
    ev = s.evaluate(jac=False, grad=True)
    out["eval_cost"], out["grad_cam"] = ev["cost"], ev["grad_cam"]
    probe = np.random.default_rng(5).normal(size=(cams0.shape[0], 9))
    ls = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR), probe=probe)
    out["Sv"], out["y_cam"], out["y_pt"], out["rhs"] = ls["Sv"], ls["y_cam"], ls["y_pt"], ls["rhs"]
    out["n_cg"], out["mcc"] = ls["n_cg"], ls["model_cost_change"]
    kw = dict(max_num_iterations=6, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    for name, o in (("it", miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, **kw)),
                    ("sj", miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, preconditioner_type=miba.PRECOND_SCHUR_JACOBI, **kw)),
                    ("ex", miba.default_options(linear_solver_type=miba.DENSE_SCHUR, **kw)),
                    ("hu", miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, loss_type=miba.LOSS_HUBER, loss_scale=1.0, **kw))):
        s.set_state(cams0, pts0)
        its, summ = s.solve(o)
        cams, pts = s.get_state(cams0, pts0)
        out[name + "_cost"] = np.array([r["cost"] for r in its])
        out[name + "_ls"] = np.array([r["linear_solver_iterations"] for r in its])
        out[name + "_ok"] = np.array([r["step_is_successful"] for r in its])
        out[name + "_final"] = summ["final_cost"]
        out[name + "_cams"], out[name + "_pts"] = cams, pts
    s.close()
    return out


if __name__ == "__main__":
    rank, world, path, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    np.savez(path, **run(rank, world, kind))

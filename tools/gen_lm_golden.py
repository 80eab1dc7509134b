#!/usr/bin/env python3
"""Generate tests/golden/lm_tiny_*.json: tiny seeded BA problems with the per-iteration LM table
computed by oracle/dense_lm.py (numpy, complex-step Jacobian, full dense normal equations) — an
implementation that shares no code with oracle/ba_oracle.c or the HIP kernels.
Run here (container) only:  python tools/gen_lm_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from bundle_adjustment_amd import synth  # noqa: E402
from oracle import dense_lm  # noqa: E402

# last field: constant-block pattern: None = all free; "reference" = what src/BundleAdjustment.cpp:77-79,88-90 marks
# (rotation and intrinsics of every camera, plus all of camera 0); "mixed" = an arbitrary pattern
CASES = [("lm_tiny_a", 8, 20, 70, 101, 0.01, 0.2, 12, None), ("lm_tiny_b", 12, 60, 260, 3, 0.01, 0.3, 15, None),
         ("lm_tiny_c", 8, 20, 70, 101, 0.01, 0.2, 12, "reference"), ("lm_tiny_d", 12, 60, 260, 3, 0.01, 0.3, 15, "mixed")]


def const_pattern(kind, n_cam):
    if kind is None:
        return None
    cc = np.zeros((n_cam, 3), bool)
    if kind == "reference":
        cc[:, 0] = True; cc[:, 2] = True; cc[0, :] = True
    else:
        rng = np.random.default_rng(77)
        cc[:] = rng.random((n_cam, 3)) < 0.3
        cc[0, :] = True
    return cc


for name, n_cam, n_pt, n_obs, seed, s_cam, s_x, n_it, ckind in CASES:
    pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
    cams0, pts0 = synth.perturb(pr, s_cam, s_x, seed=seed + 1)
    cc = const_pattern(ckind, n_cam)
    cams, pts, its, summ = dense_lm.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"], max_num_iterations=n_it,
                                          cam_const=cc)
    keys = ["iteration", "cost", "cost_change", "gradient_max_norm", "gradient_norm", "step_norm",
            "relative_decrease", "trust_region_radius", "step_is_valid", "step_is_successful"]
    out = {
        "generator": "tools/gen_lm_golden.py (oracle/dense_lm.py)",
        "n_cam": n_cam, "n_pt": n_pt, "n_obs": n_obs, "max_num_iterations": n_it,
        "cam_const": None if cc is None else cc.astype(int).tolist(),
        "cams0_hex": [float(v).hex() for v in cams0.ravel()],
        "pts0_hex": [float(v).hex() for v in pts0.ravel()],
        "obs_cam": pr["obs_cam"].tolist(), "obs_pt": pr["obs_pt"].tolist(),
        "obs_xy_hex": [float(v).hex() for v in pr["obs_xy"].ravel()],
        "iterations": [{k: (float(it[k]) if not isinstance(it[k], (bool, np.bool_)) else bool(it[k])) for k in keys} for it in its],
        "termination": summ["termination"], "final_cost": float(summ["final_cost"]),
        "cams_final_hex": [float(v).hex() for v in cams.ravel()],
        "pts_final_hex": [float(v).hex() for v in pts.ravel()],
    }
    path = os.path.join(ROOT, "tests", "golden", name + ".json")
    if os.path.exists(path) and "--force" not in sys.argv:
        print(name, "exists, kept (dense BLAS results differ in the last bits from run to run; --force rewrites)")
        continue
    with open(path, "w") as fh:
        json.dump(out, fh)
    print(name, summ, len(its))

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

CASES = [("lm_tiny_a", 8, 20, 70, 101, 0.01, 0.2, 12), ("lm_tiny_b", 12, 60, 260, 3, 0.01, 0.3, 15)]

for name, n_cam, n_pt, n_obs, seed, s_cam, s_x, n_it in CASES:
    pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
    cams0, pts0 = synth.perturb(pr, s_cam, s_x, seed=seed + 1)
    cams, pts, its, summ = dense_lm.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"], max_num_iterations=n_it)
    keys = ["iteration", "cost", "cost_change", "gradient_max_norm", "gradient_norm", "step_norm",
            "relative_decrease", "trust_region_radius", "step_is_valid", "step_is_successful"]
    out = {
        "generator": "tools/gen_lm_golden.py (oracle/dense_lm.py)",
        "n_cam": n_cam, "n_pt": n_pt, "n_obs": n_obs, "max_num_iterations": n_it,
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
    with open(path, "w") as fh:
        json.dump(out, fh)
    print(name, summ, len(its))

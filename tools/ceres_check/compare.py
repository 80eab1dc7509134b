#!/usr/bin/env python3
"""Lay Ceres' per-iteration table (tools/ceres_check/ceres_check, built only where Ceres is installed) beside the HIP path's
and the CPU oracle's on the same BAL file.  usage: compare.py build/ceres_check/ceres_check problem.txt [iterative|sparse_schur] [its]
Writes nothing; prints one line per iteration with the three costs and PCG counts and the largest relative cost difference.
Optional tooling: nothing in tests/ or bench.py needs it (there is no Ceres in the build container or on the GPU box)."""
import json, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bundle_adjustment_amd import miba
from oracle import oracle as O

exe, path = sys.argv[1], sys.argv[2]
solver = sys.argv[3] if len(sys.argv) > 3 else "iterative"
n_it = int(sys.argv[4]) if len(sys.argv) > 4 else 20
out = subprocess.run([exe, path, solver, str(n_it), str(os.cpu_count() or 1)], check=True, capture_output=True, text=True).stdout
ceres = [json.loads(l) for l in out.splitlines() if l.startswith("{\"iteration\"")]

tok = open(path).read().split()
n_cam, n_pt, n_obs = int(tok[0]), int(tok[1]), int(tok[2])
obs = np.array(tok[3:3 + 4 * n_obs], dtype=np.float64).reshape(n_obs, 4)
oc, op = obs[:, 0].astype(np.int32), obs[:, 1].astype(np.int32)
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
oxy = np.ascontiguousarray(f32(obs[:, 2:4]))
rest = f32(np.array(tok[3 + 4 * n_obs:], dtype=np.float64))
cams, pts = rest[:9 * n_cam].reshape(n_cam, 9).copy(), rest[9 * n_cam:].reshape(n_pt, 3).copy()

lst = miba.ITERATIVE_SCHUR if solver == "iterative" else miba.SPARSE_SCHUR
s = miba.Solver(device_id=0)
s.set_problem(cams, pts, oc, op, oxy)
hip, _ = s.solve(miba.default_options(linear_solver_type=lst, max_num_iterations=n_it))
s.close()
_, _, orc, _ = O.solve(cams, pts, oc, op, oxy, O.default_options(max_num_iterations=n_it, num_threads=os.cpu_count() or 1,
                       linear_solver=O.ITERATIVE_SCHUR if solver == "iterative" else O.DENSE_SCHUR))
worst = 0.0
for i, c in enumerate(ceres):
    h = hip[i] if i < len(hip) else None
    o = orc[i] if i < len(orc) else None
    print("it %3d  ceres %.10e (%d)  hip %s  oracle %s" % (c["iteration"], c["cost"], c["linear_solver_iterations"],
          "%.10e (%d)" % (h["cost"], h["linear_solver_iterations"]) if h else "-",
          "%.10e (%d)" % (o["cost"], o["linear_solver_iterations"]) if o else "-"))
    if h:
        worst = max(worst, abs(h["cost"] - c["cost"]) / c["cost"])
print("largest relative cost difference HIP vs Ceres over the common iterations: %.3e" % worst)

"""Camera model #2 on the device: the reference's cost functor as checked in (BAConstraint::operator(),
/root/reference/src/BAConstraint.cpp:15-80: inverse extrinsic, fx fy ox oy, no distortion, observed - projected), selected
with miba_problem.camera_model = MIBA_MODEL_REFERENCE.  Pinned by the 80-digit mpmath vectors of the functor's text
(tests/golden/proj2_golden.json, tools/gen_golden2.py) and compared with the oracle's restatement (bao_project2) on whole
problems: residuals and Jacobians 1e-10 of block max-norm (north_star bound; the rotation columns of cameras in Ceres'
small-angle branch, |w|^2 <= 2.2e-16, 1e-7: the device differentiates R' there, autodiff the inverse of I + [w]x), the linear
step, and LM runs with the reference's constant-block marks (src/BundleAdjustment.cpp:77-79,88-90)."""
import json
import os

import numpy as np
import pytest

from bundle_adjustment_amd import miba, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    m = np.abs(b).max()
    return np.abs(a - b).max() / (m if m > 0 else 1.0)


def kinect_problem(n_cam=12, n_pt=500, n_obs=2400, seed=3):
    """a synthetic scene seen through the reference's model: poses are camera-to-world ([R | t] maps camera to world
    coordinates, the functor inverts it), cameras look down +z, pinhole fx = fy, principal point, no distortion"""
    rng = np.random.default_rng(seed)
    pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
    w = rng.normal(0, 0.25, (n_cam, 3))
    w[0] = 0.0                                    # the reference's first camera: identity pose (small-angle branch)
    t = rng.normal(0, 0.4, (n_cam, 3)); t[0] = 0.0
    f = rng.uniform(500, 560, n_cam)
    K = np.stack([f, f, np.full(n_cam, 319.5), np.full(n_cam, 239.5), np.zeros(n_cam), np.zeros(n_cam)], 1)
    cams = np.concatenate([w, t, f[:, None], np.zeros((n_cam, 2))], 1)
    # points in front of every camera: around z = 4 in the world frame
    pts = np.stack([rng.uniform(-1.5, 1.5, n_pt), rng.uniform(-1.2, 1.2, n_pt), rng.uniform(3.0, 6.0, n_pt)], 1)
    oc, op = pr["obs_cam"], pr["obs_pt"]
    Rm = np.stack([_rot(v) for v in w])
    P = np.einsum("nji,nj->ni", Rm[oc], pts[op] - t[oc])          # R' (X - t)
    assert np.all(P[:, 2] > 0.5)
    xy = np.stack([f[oc] * P[:, 0] / P[:, 2] + 319.5, f[oc] * P[:, 1] / P[:, 2] + 239.5], 1) + rng.normal(0, 0.4, (oc.size, 2))
    xy = xy.astype(np.float32).astype(np.float64)
    cams0 = cams.copy()
    cams0[1:, 0:6] += rng.normal(0, 0.004, (n_cam - 1, 6))
    pts0 = pts + rng.normal(0, 0.03, pts.shape)
    return dict(cams0=cams0, pts0=pts0, K=K, oc=oc, op=op, xy=xy, n_cam=n_cam)


def _rot(w):
    th = np.linalg.norm(w)
    if th * th <= 2.220446049250313e-16:
        return np.array([[1, -w[2], w[1]], [w[2], 1, -w[0]], [-w[1], w[0], 1]])
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * np.outer(k, k)


def make_solver(q):
    s = miba.Solver(device_id=0)
    s.set_problem(q["cams0"], q["pts0"], q["oc"], q["op"], q["xy"], K=q["K"], camera_model=miba.MODEL_REFERENCE)
    return s


def test_functor_matches_mpmath_through_the_device():
    """every golden case as a one-camera, one-point, one-observation problem through miba_evaluate"""
    g = json.load(open(os.path.join(GOLD, "proj2_golden.json")))
    cases = [c for c in g["cases"] if float.fromhex(c["k4_hex"][0]) == float.fromhex(c["k4_hex"][1])]   # the device needs fx == fy
    assert len(cases) >= 150
    fh = lambda xs: np.array([float.fromhex(h) for h in xs])
    n = len(cases)
    cams = np.zeros((n, 9)); K = np.zeros((n, 6)); pts = np.zeros((n, 3)); xy = np.zeros((n, 2))
    for i, c in enumerate(cases):
        cams[i, 0:6] = fh(c["cam_hex"]); k4 = fh(c["k4_hex"])
        cams[i, 6] = k4[0]; K[i, 0:4] = k4; pts[i] = fh(c["X_hex"]); xy[i] = fh(c["obs_hex"])
    idx = np.arange(n, dtype=np.int32)
    s = miba.Solver(device_id=0)
    s.set_problem(cams, pts, idx, idx, xy, K=K, camera_model=miba.MODEL_REFERENCE)
    ev = s.evaluate(jac=True, grad=False)
    s.close()
    n_small = 0
    for i, c in enumerate(cases):
        R = np.array([float(v) for v in c["r"]])
        JC = np.array([[float(v) for v in row] for row in c["Jc"]])
        JP = np.array([[float(v) for v in row] for row in c["Jp"]])
        pred = max(1.0, np.abs(xy[i] - R).max())
        assert np.abs(ev["r"][i] - R).max() / pred <= 1e-10, i
        n_small += c["small_angle"]
        assert rel(ev["Jc"][i][:, 0:3], JC[:, 0:3]) <= (1e-7 if c["small_angle"] else 1e-10), (i, c["small_angle"])
        assert rel(ev["Jc"][i][:, 3:6], JC[:, 3:6]) <= 1e-10 and rel(ev["Jp"][i], JP) <= 1e-10, i
        assert np.all(ev["Jc"][i][:, 6:9] == 0.0)
    assert n_small >= 4


def test_evaluate_and_linear_step_match_oracle(oracle):
    q = kinect_problem()
    cc = np.zeros((q["n_cam"], 3), np.uint8)
    cc[0, :] = 1                          # gauge: the first camera (identity pose, small-angle branch) stays
    s = make_solver(q)
    s.set_constant_blocks(cc)
    ev = s.evaluate(jac=True, grad=True)
    probe = np.random.default_rng(8).normal(size=(q["n_cam"], 9))
    out = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, eta=1e-10, max_linear_solver_iterations=400), probe=probe)
    s.close()
    ref = oracle.evaluate(q["cams0"], q["pts0"], q["oc"], q["op"], q["xy"], cam_k4=q["K"][:, 0:4])
    assert abs(ev["cost"] - ref["cost"]) <= 1e-11 * ref["cost"]
    assert rel(ev["r"], ref["r"]) <= 1e-11 and rel(ev["Jp"], ref["Jp"]) <= 1e-10
    nz = q["oc"] != 0                     # camera 0 is in the small-angle branch (see the module docstring)
    assert rel(ev["Jc"][nz], ref["Jc"][nz]) <= 1e-10 and rel(ev["Jc"][~nz], ref["Jc"][~nz]) <= 1e-7
    assert rel(ev["grad_pt"], ref["grad_pt"]) <= 1e-10 and rel(ev["grad_cam"][1:, 0:6], ref["grad_cam"][1:, 0:6]) <= 1e-10
    # one damped linear solve: matrix-free K2 / K3 / K4 with the model's camera transforms against the oracle's stored Jacobian
    oo = oracle.default_options(linear_solver=oracle.ITERATIVE_SCHUR, eta=1e-10, max_linear_solver_iterations=400, cam_k4=q["K"][:, 0:4],
                                cam_const=cc)
    sc, sp = _oracle_scale(oracle, q)
    sc[0, :] = 0.0
    lref = oracle.linear_step(q["cams0"], q["pts0"], q["oc"], q["op"], q["xy"], 1e4, sc, sp, oo, probe=probe)
    assert lref["rc"] == 0
    assert rel(out["rhs"], lref["rhs"]) <= 1e-9 and rel(out["Sv"], lref["Sv"]) <= 1e-9
    assert out["n_cg"] == lref["n_cg"]
    assert rel(out["y_cam"], lref["y_cam"]) <= 1e-6 and rel(out["y_pt"], lref["y_pt"]) <= 1e-6
    assert abs(out["model_cost_change"] - lref["model_cost_change"]) <= 1e-8 * abs(lref["model_cost_change"])


def _oracle_scale(oracle, q):
    """Jacobi scale 1 / (1 + column norm) of the model's Jacobian, zero on the constant intrinsics"""
    ref = oracle.evaluate(q["cams0"], q["pts0"], q["oc"], q["op"], q["xy"], cam_k4=q["K"][:, 0:4])
    c2 = np.zeros((q["n_cam"], 9)); p2 = np.zeros_like(q["pts0"])
    np.add.at(c2, q["oc"], (ref["Jc"] ** 2).sum(1))
    np.add.at(p2, q["op"], (ref["Jp"] ** 2).sum(1))
    sc = 1.0 / (1.0 + np.sqrt(c2)); sc[:, 6:9] = 0.0
    return sc, 1.0 / (1.0 + np.sqrt(p2))


@pytest.mark.parametrize("marks", ["pose_free", "reference"])
@pytest.mark.parametrize("solver", ["iterative", "exact"])
def test_lm_run_matches_oracle(oracle, marks, solver):
    """LM with the model: all poses free, and the reference's own marks (rotation and intrinsics of every camera constant,
    camera 0 fixed: translations of the other cameras and the points move)"""
    if solver == "exact" and os.environ.get("MIBA_DENSE", "1") == "0":
        pytest.skip("MIBA_DENSE=0 replaces the exact solve by PCG to 1e-12: steps agree to 1e-6, not to the 1e-8 asked here")
    q = kinect_problem(seed=5)
    cc = np.zeros((q["n_cam"], 3), np.uint8)
    if marks == "reference":
        cc[:, 0] = 1; cc[:, 2] = 1; cc[0, :] = 1
    else:
        cc[0, :] = 1                      # gauge: the first camera stays
    kw = dict(max_num_iterations=8, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    s = make_solver(q)
    s.set_constant_blocks(cc)
    its, summ = s.solve(miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR if solver == "iterative" else miba.SPARSE_SCHUR, **kw))
    cams, pts = s.get_state(q["cams0"], q["pts0"])
    s.close()
    oo = oracle.default_options(linear_solver=oracle.ITERATIVE_SCHUR if solver == "iterative" else oracle.DENSE_SCHUR,
                                cam_k4=q["K"][:, 0:4], cam_const=cc, **kw)
    ocams, opts, oits, osumm = oracle.solve(q["cams0"], q["pts0"], q["oc"], q["op"], q["xy"], oo)
    n = 1
    while n < min(len(its), len(oits)) and abs(oits[n]["cost_change"]) > 1e-9 * oits[n]["cost"]:
        n += 1
    assert n >= 3
    for a, b in zip(its[:n], oits[:n]):
        assert abs(a["cost"] - b["cost"]) <= 1e-8 * b["cost"], (a, b)
        assert a["step_is_successful"] == b["step_is_successful"]
    assert abs(summ["final_cost"] - osumm["final_cost"]) <= 1e-6 * osumm["final_cost"]
    assert summ["final_cost"] < 0.5 * its[0]["cost"]
    # constant blocks did not move, intrinsics are handed back untouched
    assert np.array_equal(cams[0], q["cams0"][0]) and np.array_equal(cams[:, 6:9], q["cams0"][:, 6:9])
    if marks == "reference":
        assert np.array_equal(cams[:, 0:3], q["cams0"][:, 0:3]) and not np.array_equal(cams[1:, 3:6], q["cams0"][1:, 3:6])


def test_model_needs_equal_focal_lengths_and_keeps_the_principal_point():
    q = kinect_problem()
    s = miba.Solver(device_id=0)
    K = q["K"].copy(); K[3, 1] *= 1.01
    with pytest.raises(miba.MibaError, match="fx == fy"):
        s.set_problem(q["cams0"], q["pts0"], q["oc"], q["op"], q["xy"], K=K, camera_model=miba.MODEL_REFERENCE)
    s.set_problem(q["cams0"], q["pts0"], q["oc"], q["op"], q["xy"], K=q["K"], camera_model=miba.MODEL_REFERENCE)
    K = q["K"].copy(); K[2, 2] += 1.0
    with pytest.raises(miba.MibaError, match="principal point"):
        s.set_state(q["cams0"], q["pts0"], K=K)
    s.close()

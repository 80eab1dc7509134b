"""GPU parity tests of the exact reduced solve: explicit dense Schur complement + blocked
Cholesky on the device (csrc/miba_dense.hpp) behind the non-iterative linear_solver_types, against numpy, the
implicit operator of the same library and the oracle's BAO_DENSE_SCHUR."""
import os

import numpy as np
import pytest

from bundle_adjustment_amd import miba, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    m = np.abs(b).max()
    return np.abs(a - b).max() / (m if m > 0 else 1.0)


def make_solver(pr, cams, pts):
    s = miba.Solver(device_id=0)
    s.set_problem(cams, pts, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    return s


@pytest.fixture(scope="module")
def medium():
    pr = synth.make_problem(40, 3000, 14000, seed=11, shuffle_obs=True)
    cams0, pts0 = synth.perturb(pr, 0.01, 0.1, seed=12)
    return pr, cams0, pts0


@pytest.mark.parametrize("n", [1, 9, 63, 64, 65, 200, 777, 2000])
def test_cholesky_solve_matches_numpy(n):
    """The factorisation kernels alone (diagonal block, MFMA panel solve and trailing update, blocked
    triangular solves) on random SPD matrices, sizes around the 64-row block and beyond."""
    rng = np.random.default_rng(n)
    M = rng.normal(size=(n, n))
    A = M @ M.T + n * np.eye(n) * 0.1 + np.diag(rng.uniform(0.1, 10.0, n))
    b = rng.normal(size=n)
    s = miba.Solver(device_id=0)
    x, st = s.dense_spd_solve(A, b)
    # only the lower triangle is read
    Au = np.tril(A) + np.triu(rng.normal(size=(n, n)), 1)
    x2, st2 = s.dense_spd_solve(Au, b)
    s.close()
    ref = np.linalg.solve(A, b)
    assert st == 0 and st2 == 0
    assert rel(x, ref) <= 1e-10 * np.linalg.cond(A), (n, rel(x, ref))
    assert np.abs(A @ x - b).max() <= 1e-11 * (np.abs(A).max() * np.abs(x).max() * n)
    assert np.array_equal(x, x2)


def test_cholesky_reports_indefinite_matrix():
    n = 130
    rng = np.random.default_rng(2)
    M = rng.normal(size=(n, n))
    A = M @ M.T
    A[100, 100] = -1.0
    s = miba.Solver(device_id=0)
    _, st = s.dense_spd_solve(A, np.ones(n))
    s.close()
    assert st == 2


@pytest.fixture(autouse=True)
def dense_path():
    """The tiles (tests/test_gpu_sparse.py) are the default exact solver at every size; this file keeps the dense
    factorisation, their fallback, covered."""
    old = os.environ.get("MIBA_EXACT")
    os.environ["MIBA_EXACT"] = "dense"
    yield
    if old is None:
        os.environ.pop("MIBA_EXACT", None)
    else:
        os.environ["MIBA_EXACT"] = old


@pytest.fixture(params=["planned", "planned_host", "atomics"])
def assembly(request):
    """Both assembly kernels: the planned, atomic-free k_dense_blocks (default; instance list sorted on the device, or by the
    host threads: MIBA_DENSE_PLAN=host) and k_dense_pairs / k_dense_pairs_giant (global atomics, used when the pair plan
    would be too large)."""
    old = os.environ.get("MIBA_DENSE_PLAN")
    os.environ["MIBA_DENSE_PLAN"] = {"atomics": "0", "planned_host": "host", "planned": "1"}[request.param]
    yield request.param
    if old is None:
        os.environ.pop("MIBA_DENSE_PLAN", None)
    else:
        os.environ["MIBA_DENSE_PLAN"] = old


def test_reduced_system_matches_implicit_operator(medium, assembly):
    """S assembled explicitly (+ k_dense_diag) against the matrix-free product S v of the same
    state, and S y = rhs for the exact step."""
    pr, cams0, pts0 = medium
    rng = np.random.default_rng(7)
    s = make_solver(pr, cams0, pts0)
    opt = miba.default_options(linear_solver_type=miba.DENSE_SCHUR)
    n = 9 * pr["n_cam"]
    worst = 0.0
    out = None
    S_first = None
    for _ in range(4):
        probe = rng.normal(size=(pr["n_cam"], 9))
        out = s.linear_step(1e4, opt, probe=probe)
        S = s.reduced_system()
        assert np.array_equal(S, S.T)
        if assembly != "atomics":   # no atomics: the same bits every time
            assert S_first is None or np.array_equal(S, S_first)
            S_first = S
        worst = max(worst, rel(S @ probe.ravel(), out["Sv"].ravel()))
    assert worst <= 1e-11, worst
    assert out["n_cg"] == 1
    # exact step: residual of the reduced system at rounding level
    y = out["y_cam"].ravel()
    r = S @ y - out["rhs"].ravel()
    assert np.abs(r).max() <= 1e-9 * np.abs(out["rhs"]).max(), np.abs(r).max()
    assert rel(y, np.linalg.solve(S, out["rhs"].ravel())) <= 1e-7
    # and the PCG path of the same library converges to it
    it = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, eta=1e-15,
                                                 max_linear_solver_iterations=5000))
    s.close()
    assert rel(it["y_cam"], out["y_cam"]) <= 1e-5
    assert abs(it["model_cost_change"] - out["model_cost_change"]) <= 1e-8 * abs(out["model_cost_change"])


def test_device_plan_is_the_host_plan(medium):
    """The instance list sorted on the device and the one counting-sorted by the host threads are the same list: S comes
    out bit for bit the same, on a problem with a long track too (the workgroup-per-point fill)."""
    rng = np.random.default_rng(3)
    pr = synth.make_problem(150, 400, 2500, seed=31)
    pts = np.concatenate([pr["pts"], np.array([[0.2, -0.1, 0.3]])])
    cs = rng.choice(150, size=140, replace=False).astype(np.int32)
    ps = np.full(140, 400, dtype=np.int32)
    xy, z = synth.project(pr["cams"], pts, cs, ps)
    ok = z < 0
    assert ok.sum() > 64
    q = dict(pr)
    q["n_pt"] = 401
    q["obs_cam"] = np.concatenate([pr["obs_cam"], cs[ok]]); q["obs_pt"] = np.concatenate([pr["obs_pt"], ps[ok]])
    q["obs_xy"] = np.concatenate([pr["obs_xy"], xy[ok] + rng.normal(0, 0.5, xy[ok].shape)])
    cams_l = pr["cams"].copy(); cams_l[:, 0:6] += rng.normal(0, 0.003, (150, 6))
    pts_l = pts + rng.normal(0, 0.05, pts.shape)
    for prob in (medium, (q, cams_l, pts_l)):
        pr_, cams0, pts0 = prob
        mats = []
        for plan in ("1", "host"):
            old = os.environ.get("MIBA_DENSE_PLAN")
            os.environ["MIBA_DENSE_PLAN"] = plan
            try:
                s = make_solver(pr_, cams0, pts0)
                s.linear_step(1e4, miba.default_options(linear_solver_type=miba.DENSE_SCHUR))
                mats.append(s.reduced_system())
                s.close()
            finally:
                if old is None:
                    os.environ.pop("MIBA_DENSE_PLAN", None)
                else:
                    os.environ["MIBA_DENSE_PLAN"] = old
        assert np.array_equal(mats[0], mats[1])


def test_planned_assembly_with_a_duplicated_observation(medium):
    """A point that sees one camera twice: the device plan steps aside for the host's (which counts such pairs from both
    sides), and the planned assembly still agrees with the atomics kernel and with the tiles."""
    pr, cams0, pts0 = medium
    q = dict(pr)
    q["obs_cam"] = np.concatenate([pr["obs_cam"], pr["obs_cam"][5:6]])
    q["obs_pt"] = np.concatenate([pr["obs_pt"], pr["obs_pt"][5:6]])
    q["obs_xy"] = np.concatenate([pr["obs_xy"], pr["obs_xy"][5:6] + 0.25])
    outs = {}
    for label, env_kw in (("planned", {"MIBA_DENSE_PLAN": "1"}), ("atomics", {"MIBA_DENSE_PLAN": "0"}),
                          ("tiles", {"MIBA_DENSE_PLAN": "1", "MIBA_EXACT": "sparse"})):
        old = {k: os.environ.get(k) for k in env_kw}
        os.environ.update(env_kw)
        try:
            s = make_solver(q, cams0, pts0)
            outs[label] = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.DENSE_SCHUR))
            s.close()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    for label in ("atomics", "tiles"):
        assert outs[label]["n_cg"] == 1
        assert rel(outs[label]["y_cam"], outs["planned"]["y_cam"]) <= 1e-9, label
        assert rel(outs[label]["y_pt"], outs["planned"]["y_pt"]) <= 1e-9, label


def test_exact_step_matches_oracle_dense_schur(oracle, medium):
    pr, cams0, pts0 = medium
    s = make_solver(pr, cams0, pts0)
    out = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.SPARSE_NORMAL_CHOLESKY))
    s.close()
    sc, sp = oracle.jacobi_scale(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    ref = oracle.linear_step(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"], 1e4, sc, sp,
                             oracle.default_options(linear_solver=oracle.DENSE_SCHUR))
    assert ref["rc"] == 0
    assert rel(out["rhs"], ref["rhs"]) <= 1e-10
    assert rel(out["y_cam"], ref["y_cam"]) <= 1e-7
    assert rel(out["y_pt"], ref["y_pt"]) <= 1e-7
    assert abs(out["model_cost_change"] - ref["model_cost_change"]) <= 1e-10 * abs(ref["model_cost_change"])


def test_dense_path_with_giant_points_and_constant_blocks(oracle, assembly):
    """Tracks longer than a wavelet (k_dense_pairs_giant), duplicate-free ordinary points, and constant
    parameter blocks (zero scale: their rows of S keep only the LM diagonal)."""
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
    assert ok.sum() > 64
    cams0 = cams.copy(); cams0[:, 0:6] += rng.normal(0, 0.003, (150, 6))
    pts0 = pts + rng.normal(0, 0.05, pts.shape)
    const = np.zeros((150, 3), dtype=np.uint8)
    const[0, :] = 1; const[5, 2] = 1; const[9, 0] = 1
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, oc, op, oxy)
    s.set_constant_blocks(const)
    probe = rng.normal(size=(150, 9))
    out = s.linear_step(1e3, miba.default_options(linear_solver_type=miba.DENSE_SCHUR), probe=probe)
    S = s.reduced_system()
    s.close()
    assert rel(S @ probe.ravel(), out["Sv"].ravel()) <= 1e-11
    assert np.abs(S @ out["y_cam"].ravel() - out["rhs"].ravel()).max() <= 1e-9 * np.abs(out["rhs"]).max()
    assert np.all(out["y_cam"][0] == 0.0) and np.all(out["y_cam"][5, 6:9] == 0.0) and np.all(out["y_cam"][9, 0:3] == 0.0)
    sc, sp = oracle.jacobi_scale(cams0, pts0, oc, op, oxy)
    sc = sc * (1.0 - np.repeat(const, 3, axis=1))   # a zero scale takes the column out (as bao_solve does)
    ref = oracle.linear_step(cams0, pts0, oc, op, oxy, 1e3, sc, sp,
                             oracle.default_options(linear_solver=oracle.DENSE_SCHUR))
    assert rel(out["y_cam"], ref["y_cam"]) <= 1e-7
    assert rel(out["y_pt"], ref["y_pt"]) <= 1e-7


def test_dense_env_switch_falls_back_to_pcg(medium):
    pr, cams0, pts0 = medium
    os.environ["MIBA_DENSE"] = "0"
    try:
        s = make_solver(pr, cams0, pts0)
        out = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.DENSE_SCHUR))
        s.close()
    finally:
        os.environ.pop("MIBA_DENSE", None)
    assert out["n_cg"] > 1


def test_schur_jacobi_preconditioner_matches_oracle(oracle, medium):
    """ceres::SCHUR_JACOBI (block diagonal of S itself as the PCG preconditioner; k_mf_schurdiag): same PCG
    recurrence as the oracle run with the same preconditioner => same iteration counts and steps; and the
    preconditioner blocks are the inverse diagonal blocks of the explicit S."""
    pr, cams0, pts0 = medium
    sc, sp = oracle.jacobi_scale(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    s = make_solver(pr, cams0, pts0)
    for pc_m, pc_o in ((miba.PRECOND_JACOBI, oracle.JACOBI), (miba.PRECOND_SCHUR_JACOBI, oracle.SCHUR_JACOBI)):
        for eta in (0.1, 1e-6):
            out = s.linear_step(1e4, miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, eta=eta, preconditioner_type=pc_m))
            ref = oracle.linear_step(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"], 1e4, sc, sp,
                                     oracle.default_options(linear_solver=oracle.ITERATIVE_SCHUR, eta=eta, preconditioner=pc_o))
            assert out["n_cg"] == ref["n_cg"], (pc_m, eta, out["n_cg"], ref["n_cg"])
            assert rel(out["y_cam"], ref["y_cam"]) <= 1e-7
            assert abs(out["model_cost_change"] - ref["model_cost_change"]) <= 1e-9 * abs(ref["model_cost_change"])
    s.close()


def test_schur_jacobi_lm_run_matches_oracle(oracle, medium):
    pr, cams0, pts0 = medium
    kw = dict(max_num_iterations=8, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    s = make_solver(pr, cams0, pts0)
    its, summ = s.solve(miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, preconditioner_type=miba.PRECOND_SCHUR_JACOBI, **kw))
    s.close()
    _, _, oits, osumm = oracle.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                                     oracle.default_options(linear_solver=oracle.ITERATIVE_SCHUR, preconditioner=oracle.SCHUR_JACOBI, **kw))
    n_cmp = 1
    while n_cmp < len(oits) and abs(oits[n_cmp]["cost_change"]) > 1e-9 * oits[n_cmp]["cost"]:
        n_cmp += 1
    assert n_cmp >= 4
    for a, b in zip(its[:n_cmp], oits[:n_cmp]):
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * abs(b["cost"]), (a, b)
        assert a["linear_solver_iterations"] == b["linear_solver_iterations"]
    assert abs(summ["final_cost"] - osumm["final_cost"]) <= 1e-6 * osumm["final_cost"]

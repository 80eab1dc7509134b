"""Edge cases of the C ABI on the GPU: empty and degenerate problems, invalid indices, single-observation points,
cameras and points that no observation references (src/BundleAdjustment.cpp:49-60 leaves them out of the problem)."""
import numpy as np
import pytest

from bundle_adjustment_amd import miba, synth

pytestmark = pytest.mark.gpu


def _tiny():
    pr = synth.make_problem(6, 40, 150, seed=31)
    cams0, pts0 = synth.perturb(pr, 0.005, 0.05, seed=32)
    return pr, cams0, pts0


def test_no_observations():
    pr, cams0, pts0 = _tiny()
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))
    assert s.cost() == 0.0
    its, summ = s.solve(miba.default_options(max_num_iterations=3))
    cams, pts = s.get_state(cams0, pts0)
    s.close()
    assert summ["final_cost"] == 0.0 and len(its) >= 1
    assert np.array_equal(cams, cams0) and np.array_equal(pts, pts0)


def test_every_observation_filtered_out():
    pr, cams0, pts0 = _tiny()
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, pr["obs_cam"], np.full_like(pr["obs_pt"], -1), pr["obs_xy"])
    its, summ = s.solve(miba.default_options(max_num_iterations=3))
    s.close()
    assert summ["final_cost"] == 0.0 and summ["num_residual_blocks"] == 0


def test_out_of_range_index_is_an_error():
    pr, cams0, pts0 = _tiny()
    oc = pr["obs_cam"].copy()
    oc[3] = 6
    s = miba.Solver(device_id=0)
    with pytest.raises(miba.MibaError):
        s.set_problem(cams0, pts0, oc, pr["obs_pt"], pr["obs_xy"])
    op = pr["obs_pt"].copy()
    op[7] = 40
    with pytest.raises(miba.MibaError):
        s.set_problem(cams0, pts0, pr["obs_cam"], op, pr["obs_xy"])
    # the handle is still usable
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    assert s.cost() > 0
    s.close()


def test_solve_before_set_problem_is_an_error():
    s = miba.Solver(device_id=0)
    with pytest.raises(miba.MibaError):
        s.solve(miba.default_options())
    s.close()


def test_unreferenced_blocks_and_single_observation_points(oracle):
    """Two extra cameras and three extra points that nothing observes, plus points seen once: they stay where they
    are (unreferenced) or move along with the rest (single observation), exactly as in the oracle."""
    pr, cams0, pts0 = _tiny()
    rng = np.random.default_rng(5)
    cams = np.concatenate([cams0, cams0[:2] + 0.1])
    pts = np.concatenate([pts0, rng.normal(size=(3, 3)), rng.normal(size=(4, 3)) * 0.5])
    # four new points (ids 43..46) with one observation each
    n0 = pts0.shape[0] + 3
    oc = np.concatenate([pr["obs_cam"], np.array([0, 1, 2, 3], np.int32)])
    op = np.concatenate([pr["obs_pt"], np.arange(n0, n0 + 4, dtype=np.int32)])
    xy, _ = synth.project(cams, pts, oc[-4:], op[-4:])
    oxy = np.concatenate([pr["obs_xy"], xy + 0.3])
    s = miba.Solver(device_id=0)
    s.set_problem(cams, pts, oc, op, oxy)
    kw = dict(max_num_iterations=6, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    its, summ = s.solve(miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, **kw))
    c1, p1 = s.get_state(cams, pts)
    s.close()
    c2, p2, oits, osumm = oracle.solve(cams, pts, oc, op, oxy, oracle.default_options(linear_solver=oracle.ITERATIVE_SCHUR, **kw))
    assert np.array_equal(c1[6:], cams[6:]) and np.array_equal(p1[40:43], pts[40:43])
    assert not np.array_equal(p1[43:], pts[43:])
    for a, b in zip(its, oits):
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * max(b["cost"], 1e-30)
    assert np.abs(c1 - c2).max() <= 1e-7 and np.abs(p1 - p2).max() <= 1e-6
    assert summ["num_active_cameras"] == 6 and summ["num_active_points"] == 44


@pytest.mark.parametrize("solver", ["iterative", "exact", "huber"])
def test_run_to_run_reproducibility(solver):
    """The GPU race check SURVEY.md §5 asks for: the same solve three times (twice on one handle, once on a fresh one)
    must tell the same story.  The per-tile camera sums are LDS floating-point atomics, so the runs agree to summation
    order (1e-9 on costs: a PCG run to a tight tolerance amplifies the last bits), not bit for bit; a race would show up as a different PCG count, a different accept/reject
    decision or a cost off by far more than that.  The residual/Jacobian evaluation has no atomics: bit-identical."""
    pr = synth.make_problem(60, 6000, 30000, seed=41, shuffle_obs=True)
    cams0, pts0 = synth.perturb(pr, 0.005, 0.05, seed=42)
    kw = dict(max_num_iterations=6, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    opt = {"iterative": miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, **kw),
           "exact": miba.default_options(linear_solver_type=miba.SPARSE_SCHUR, **kw),
           "huber": miba.default_options(linear_solver_type=miba.ITERATIVE_SCHUR, loss_type=miba.LOSS_HUBER, loss_scale=1.0, **kw)}[solver]
    runs, evals = [], []
    for fresh in range(2):
        s = miba.Solver(device_id=0)
        s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
        for _ in range(2 - fresh):
            s.set_state(cams0, pts0)
            evals.append(s.evaluate(jac=True, grad=False))
            try:
                runs.append(s.solve(opt))
            except miba.MibaError:
                s.close()
                if solver == "huber":
                    pytest.skip("the loss needs the two-pass matrix-free kernels")
                raise
        s.close()
    assert len(runs) == 3
    for ev in evals[1:]:
        assert np.array_equal(ev["r"], evals[0]["r"]) and np.array_equal(ev["Jc"], evals[0]["Jc"]) and ev["cost"] == evals[0]["cost"]
    its0, summ0 = runs[0]
    for its, summ in runs[1:]:
        assert len(its) == len(its0)
        for a, b in zip(its, its0):
            assert a["step_is_successful"] == b["step_is_successful"]
            # (MIBA_DENSE=0 replaces the exact solve by PCG to q-tolerance 1e-12, whose stopping test sits in
            # rounding noise: its length wanders by a few iterations from run to run)
            if solver != "exact" or b["linear_solver_iterations"] <= 1:
                assert a["linear_solver_iterations"] == b["linear_solver_iterations"]
            assert abs(a["cost"] - b["cost"]) <= 1e-9 * b["cost"]
        assert abs(summ["final_cost"] - summ0["final_cost"]) <= 1e-9 * summ0["final_cost"]


def test_intrinsics_that_are_not_bal_shaped_are_refused():
    """The device camera has one focal length and no principal point (src/BalDataloader.cpp:143-148); fy != fx or a
    non-zero (ox, oy) is an error, not something miba_get_state silently overwrites."""
    pr, cams0, pts0 = _tiny()
    _, _, K = miba.split_cams(cams0)
    s = miba.Solver(device_id=0)
    for col, val in ((1, 1.0), (2, 3.5), (3, -2.0)):
        Kb = K.copy()
        Kb[2, col] += val
        with pytest.raises(miba.MibaError):
            s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"], K=Kb)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    with pytest.raises(miba.MibaError):
        Kb = K.copy()
        Kb[0, 3] = 1.0
        s.set_state(cams0, pts0, K=Kb)
    cams, _ = s.get_state(cams0, pts0)
    s.close()
    assert np.array_equal(cams, cams0)


def test_run_that_ends_on_function_tolerance_has_ceres_record_count(oracle):
    """TrustRegionMinimizer returns from the parameter / function tolerance tests without appending the unfinished
    iteration to summary.iterations or counting a step: same records and counts as the oracle."""
    pr, cams0, pts0 = _tiny()
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    its, summ = s.solve(miba.default_options(linear_solver_type=miba.SPARSE_SCHUR, max_num_iterations=50))
    s.close()
    _, _, oits, osumm = oracle.solve(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                                     oracle.default_options(max_num_iterations=50, linear_solver=oracle.DENSE_SCHUR))
    assert summ["termination_type"] == miba.CONVERGENCE and "tolerance" in summ["message"]
    assert len(its) == len(oits) == osumm["num_iterations"]
    assert its[-1]["iteration"] == len(its) - 1
    assert summ["num_iterations"] == len(its) - 1 == summ["num_successful_steps"] + summ["num_unsuccessful_steps"]
    assert summ["num_successful_steps"] == osumm["num_successful"] and summ["num_unsuccessful_steps"] == osumm["num_unsuccessful"]


def _lm_table(pr, cams0, pts0, iters=6):
    s = miba.Solver(device_id=0)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    its, summ = s.solve(miba.default_options(max_num_iterations=iters, function_tolerance=0.0, gradient_tolerance=0.0,
                                             parameter_tolerance=0.0))
    s.close()
    return [(r["step_is_successful"], r["linear_solver_iterations"]) for r in its], [r["cost"] for r in its], summ


def _same_run(a, b):
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[1], b[1]):
        assert abs(x - y) <= 1e-9 * abs(y)


@pytest.fixture(scope="module")
def tail_problem():
    pr = synth.make_problem(300, 20000, 120000, seed=31)
    cams0, pts0 = synth.perturb(pr, 0.005, 0.05, seed=32)
    return pr, cams0, pts0


def test_pcg_tail_without_grid_barriers(tail_problem, monkeypatch):
    """MIBA_TAIL_SPLIT=1: phase A, phase B and phase C of the PCG tail as three launches, no grid barrier (the form for a device
    that this solver does not have to itself): the same LM table as the one-launch tail."""
    pr, cams0, pts0 = tail_problem
    ref = _lm_table(pr, cams0, pts0)
    monkeypatch.setenv("MIBA_TAIL_SPLIT", "1")
    _same_run(_lm_table(pr, cams0, pts0), ref)


def test_pcg_tail_barrier_timeout_falls_back(tail_problem, monkeypatch):
    """A grid barrier that times out (here: a patience of one tick of the 100 MHz counter, which no barrier of several
    workgroups meets) no longer ends the solve with MIBA_ERR_HIP: the handle switches to the tail without barriers, the
    linear solve starts over, and the LM table is the one of the undisturbed run."""
    pr, cams0, pts0 = tail_problem
    ref = _lm_table(pr, cams0, pts0)
    monkeypatch.setenv("MIBA_TAIL_TIMEOUT_TICKS", "1")
    _same_run(_lm_table(pr, cams0, pts0), ref)


def test_solve_while_another_stream_keeps_the_device_busy(tail_problem):
    """The PCG tail's one-launch form needs its grid resident as a whole.  A second handle of the same process (its own stream)
    that keeps the compute units busy with back-to-back products of a two-million-observation problem delays it but must not
    break it: same LM table, no error (a barrier that did time out would switch the handle to the tail without barriers)."""
    import threading
    pr, cams0, pts0 = tail_problem
    ref = _lm_table(pr, cams0, pts0)
    big = synth.make_problem(2000, 300000, 2000000, seed=77)
    bc, bp = synth.perturb(big, 0.002, 0.03, seed=78)
    hog = miba.Solver(device_id=0)
    hog.set_problem(bc, bp, big["obs_cam"], big["obs_pt"], big["obs_xy"])
    probe = np.random.default_rng(1).normal(size=(big["n_cam"], 9))
    hog.linear_step(1e4, miba.default_options(max_linear_solver_iterations=1), probe=probe)
    stop = threading.Event()
    busy = []

    def run_hog():
        while not stop.is_set():
            busy.append(hog.time_matvec(probe, 50))

    t = threading.Thread(target=run_hog, daemon=True)
    t.start()
    try:
        got = _lm_table(pr, cams0, pts0)
    finally:
        stop.set()
        t.join(timeout=120)
        hog.close()
    assert busy, "the second handle never ran"
    _same_run(got, ref)


def test_kernel_times_report_the_shape_of_a_pcg_iteration_and_the_product_clock():
    """measurement fields of miba_get_kernel_times (bench.py prices the roofline with them): launches per PCG iteration, how q
    crossed the ranks, and the shader clock the camera-major product kernel measured over one workgroup's life"""
    pr = synth.make_problem(300, 30000, 200000, seed=77)
    cams0, pts0 = synth.perturb(pr, 0.003, 0.05, seed=78)
    import os
    old = os.environ.get("MIBA_K3")
    os.environ["MIBA_K3"] = "cm"
    try:
        s = miba.Solver(device_id=0)
        s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
        s.solve(miba.default_options(max_num_iterations=2, linear_solver_type=miba.ITERATIVE_SCHUR))
        kt = s.kernel_times()
        s.close()
    finally:
        if old is None:
            os.environ.pop("MIBA_K3", None)
        else:
            os.environ["MIBA_K3"] = old
    # (the environment matrix of tools/gpu_env_matrix.sh runs this under the knobs that change the shape)
    forced = os.environ.get("MIBA_FORCE_RCCL") == "1"
    split = os.environ.get("MIBA_TAIL_SPLIT") == "1"
    assert kt["pcg_launches"] == (4 if split else 3 if forced else 2) and kt["pcg_exchange"] == (1 if forced else 0)
    assert kt["product_layout"] == 1
    assert 500.0 < kt["product_clock_mhz"] < 3000.0, kt["product_clock_mhz"]

"""Iteration callbacks beside the solver (miba_options.async_callbacks; the reference's update_state_every_iteration + callbacks,
src/BundleAdjustment.cpp:104-105, src/MeshWriterCallback.h:17-23, off the solver's critical path): the callback runs on a
worker thread from pinned state snapshots while the device works on the next iteration.  What it sees must be what the
synchronous callback sees: the same iteration records and, through miba_get_state, bit for bit the same states."""
import threading

import numpy as np
import pytest

from bundle_adjustment_amd import miba, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prob():
    pr = synth.make_problem(60, 6000, 30000, seed=41, shuffle_obs=True)
    cams0, pts0 = synth.perturb(pr, 0.01, 0.1, seed=42)
    return pr, cams0, pts0


def run(prob, async_cb, devices=None, stop_at=None, verdict=1):
    pr, cams0, pts0 = prob
    s = miba.Solver(device_id=0) if devices is None else miba.Solver(devices=devices)
    s.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    seen = []
    threads = set()

    def cb(it):
        threads.add(threading.get_ident())
        cams, pts = s.get_state(cams0, pts0)
        seen.append((it.iteration, it.cost, it.step_is_successful, cams.copy(), pts.copy()))
        if stop_at is not None and it.iteration == stop_at:
            return verdict
        return 0

    opt = miba.default_options(max_num_iterations=8, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    opt.async_callbacks = 1 if async_cb else 0
    its, summ = s.solve(opt, callback=cb)
    final = s.get_state(cams0, pts0)
    s.close()
    return its, summ, seen, threads, final


@pytest.mark.parametrize("devices", [None, [0, 0, 0]])
def test_async_callback_sees_the_same_states(prob, devices):
    pr, cams0, pts0 = prob
    a = run(prob, False, devices)
    b = run(prob, True, devices)
    assert b[3] and threading.get_ident() not in b[3]          # the callback ran on the solve's worker thread
    assert threading.get_ident() in a[3]
    assert len(a[2]) == len(b[2]) == len(a[0]) == len(b[0])
    # the same run as with the callback on the calling thread (up to the last-bit noise of two runs: LDS atomics)
    for x, y in zip(a[2], b[2]):
        assert x[0] == y[0] and x[2] == y[2]
        assert abs(x[1] - y[1]) <= 1e-12 * abs(x[1])
        assert np.abs(x[3] - y[3]).max() <= 1e-9 and np.abs(x[4] - y[4]).max() <= 1e-9
    # every callback saw the state of ITS iteration: the cost of the state it was handed is the cost of its record
    # (accepted steps; a rejected step's record carries the candidate's cost, the state stays the last accepted one)
    chk = miba.Solver(device_id=0)
    chk.set_problem(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
    last = None
    for k, (itn, cost, ok, cams, pts) in enumerate(b[2]):
        chk.set_state(cams, pts)
        c = chk.cost()
        if k == 0 or ok:
            assert abs(c - cost) <= 1e-12 * cost, (itn, c, cost)
            last = c
        else:
            assert abs(c - last) <= 1e-12 * last
    chk.close()
    # states change from iteration to iteration exactly where steps were accepted (the snapshots are not one buffer seen late)
    changed = [not np.array_equal(b[2][i][4], b[2][i - 1][4]) for i in range(1, len(b[2]))]
    assert changed == [bool(r[2]) for r in b[2][1:]]
    assert np.abs(a[4][1] - b[4][1]).max() <= 1e-9
    assert abs(a[1]["final_cost"] - b[1]["final_cost"]) <= 1e-12 * a[1]["final_cost"]


@pytest.mark.parametrize("verdict,kind", [(1, miba.USER_FAILURE), (2, miba.USER_SUCCESS)])
def test_async_stop_takes_effect_one_iteration_late(prob, verdict, kind):
    a = run(prob, False, stop_at=2, verdict=verdict)
    b = run(prob, True, stop_at=2, verdict=verdict)
    assert a[1]["termination_type"] == kind and b[1]["termination_type"] == kind
    assert len(a[0]) == 3            # iterations 0, 1, 2
    assert len(b[0]) == 4            # the verdict of iteration 2 arrives while iteration 3 is being computed
    assert [x[0] for x in b[2]] == [0, 1, 2]   # no callback after the one that asked to stop

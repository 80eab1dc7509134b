"""Multi-rank control flow on ONE GPU: WORLD processes, each a rank of libmiba with its own shard of the points,
their all-reduces staged through the shared-memory TEST transport (MIBA_TEST_SHM_COMM; RCCL refuses two ranks on
one device).  Every rank must report what the one-rank run reports: the same PCG iteration counts, accept/reject
decisions, costs (to summation-order rounding) and final state, for the iterative, SCHUR_JACOBI and exact
solvers and for the iterative solver with a Huber loss.  What this cannot cover is RCCL itself and xGMI; test_rccl_path_single_rank covers the RCCL calls."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "_multirank_worker.py")


def launch(world, tmp_path, kind="medium", extra_env=None):
    name = "miba_test_%d_%d_%s" % (os.getpid(), world, kind)
    env = dict(os.environ, MIBA_TEST_SHM_COMM=name, **(extra_env or {}))
    env.pop("MIBA_TEST_NO_COMM", None)
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(tmp_path / ("w%d_r%d.npz" % (world, r))), kind],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=150)
            outs.append(o)
    finally:
        for p in procs:           # a rank that died leaves the others spinning in a barrier: end them by PID
            if p.poll() is None:
                p.kill()
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d of %d failed:\n%s" % (r, world, outs[r][-3000:])
    return [dict(np.load(str(tmp_path / ("w%d_r%d.npz" % (world, r))))) for r in range(world)]


@pytest.fixture(scope="module")
def single(tmp_path_factory):
    return launch(1, tmp_path_factory.mktemp("w1"))[0]


@pytest.fixture(scope="module")
def single_giant(tmp_path_factory):
    return launch(1, tmp_path_factory.mktemp("w1g"), "giant", {"MIBA_TILE_CHUNKS": "1"})[0]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_agree_with_single_rank(single, world, tmp_path):
    check(launch(world, tmp_path), single, world)


def test_ranks_agree_with_single_rank_empty_shards(single, tmp_path):
    """Eight ranks for a problem of seven default-size tiles: at least one rank owns nothing and still has to take
    part in every reduction with zeros."""
    ranks = launch(8, tmp_path, "medium", {"MIBA_TILE_CHUNKS": "8"})
    assert any(np.abs(o["y_pt"]).sum() == 0.0 for o in ranks)
    check(ranks, single, 8)


def test_ranks_agree_with_single_rank_giant_points(single_giant, tmp_path):
    """One-chunk tiles (so that four ranks all own tiles) and a point with more than 64 observations."""
    check(launch(4, tmp_path, "giant", {"MIBA_TILE_CHUNKS": "1"}), single_giant, 4)


def check(ranks, ref, world):
    for r, out in enumerate(ranks):
        # every rank holds the same replicated results, bit for bit (sums combined in rank order everywhere)
        for k in out:
            if k != "y_pt":   # the point step is handed back by the rank that owns the point (zeros elsewhere)
                assert np.array_equal(out[k], ranks[0][k]), (world, r, k)
    owners = sum((np.abs(o["y_pt"]).sum(1) > 0).astype(int) for o in ranks)
    assert owners.max() == 1   # shards are disjoint
    out = dict(ranks[0])
    out["y_pt"] = sum(o["y_pt"] for o in ranks)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert rel(out["rhs"], ref["rhs"]) <= 1e-11 and rel(out["Sv"], ref["Sv"]) <= 1e-11
    assert int(out["n_cg"]) == int(ref["n_cg"]) and rel(out["y_cam"], ref["y_cam"]) <= 1e-8 and rel(out["y_pt"], ref["y_pt"]) <= 1e-8
    assert abs(out["mcc"] - ref["mcc"]) <= 1e-10 * abs(ref["mcc"])
    assert abs(out["eval_cost"] - ref["eval_cost"]) <= 1e-12 * ref["eval_cost"] and rel(out["grad_cam"], ref["grad_cam"]) <= 1e-10
    for name in ("it", "sj", "ex", "hu"):   # iterative, SCHUR_JACOBI, exact, iterative with a Huber loss
        # iterations are compared while the one-rank run still makes progress: beyond that, accept / reject is
        # decided by rounding noise (same rule as test_lm_trajectory_matches_oracle)
        c = ref[name + "_cost"]
        n = 1
        while n < min(len(c), len(out[name + "_cost"])) and abs(c[n] - c[n - 1]) > 1e-9 * c[n]:
            n += 1
        assert n >= 4, (name, c)
        assert np.array_equal(out[name + "_ok"][:n], ref[name + "_ok"][:n]), name
        assert np.abs(out[name + "_cost"][:n] - ref[name + "_cost"][:n]).max() <= 1e-8 * ref[name + "_cost"][0], name
        if name != "ex":
            assert np.abs(out[name + "_ls"][:n] - ref[name + "_ls"][:n]).max() <= 1, (name, out[name + "_ls"], ref[name + "_ls"])
        assert abs(out[name + "_final"] - ref[name + "_final"]) <= 1e-8 * ref[name + "_final"], name
        # states: looser than costs (inexact PCG steps move along weakly determined directions at rounding level)
        assert rel(out[name + "_pts"], ref[name + "_pts"]) <= 1e-4 and rel(out[name + "_cams"], ref[name + "_cams"]) <= 1e-4, name

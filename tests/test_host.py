"""CPU tests of the host side of libmiba: the C-ABI library loads and exports every symbol that
include/miba.h declares, fails loudly without a GPU, and the tiler / sharder (miba_plan_*) keeps its
invariants.  No compute entry point is called here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from bundle_adjustment_amd import miba, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "miba.h")).read()
    declared = set(re.findall(r"\b(miba_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"miba_iteration_callback"}
    assert declared == set(miba.EXPORTS), declared ^ set(miba.EXPORTS)
    lib = miba.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(miba.MibaError) as e:
        miba.Solver()
    assert "no CPU fallback" in str(e.value)


def test_default_options_are_ceres_defaults():
    o = miba.default_options()
    assert (o.max_num_iterations, o.linear_solver_type, o.jacobi_scaling) == (50, miba.ITERATIVE_SCHUR, 1)
    assert (o.max_linear_solver_iterations, o.min_linear_solver_iterations, o.max_num_consecutive_invalid_steps) == (500, 0, 5)
    assert (o.initial_trust_region_radius, o.max_trust_region_radius, o.min_trust_region_radius) == (1e4, 1e16, 1e-32)
    assert (o.min_relative_decrease, o.min_lm_diagonal, o.max_lm_diagonal) == (1e-3, 1e-6, 1e32)
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance, o.eta) == (1e-6, 1e-10, 1e-8, 1e-1)


@pytest.mark.parametrize("world", [1, 2, 8])
def test_plan_invariants(world):
    pr = synth.make_problem(60, 5000, 26000, seed=2, shuffle_obs=True)
    oc, op, oxy = pr["obs_cam"].copy(), pr["obs_pt"].copy(), pr["obs_xy"].copy()
    op[3] = -1
    oxy[10, 0] = np.nan
    info = miba.plan_shards(60, 5000, oc, op, oxy, world)
    assert info["n_valid"] == 26000 - 2
    assert info["n_slots"] % 256 == 0 and info["n_slots"] >= info["n_valid"]
    assert info["n_slots"] <= 1.25 * info["n_valid"] + 256 * 8        # padding stays small
    assert sum(info["tiles_per_rank"]) == info["n_tiles"]
    assert sum(info["obs_per_rank"]) == info["n_valid"]
    mean = info["n_valid"] / world
    assert max(info["obs_per_rank"]) <= 1.1 * mean + 1024              # shards balanced by observations
    owner = miba.plan_point_owner(60, 5000, oc, op, oxy, world)
    assert owner.min() >= 0 and owner.max() == world - 1
    keep = (op != -1) & ~np.isnan(oxy[:, 0])
    per_rank = np.bincount(owner[op[keep]], minlength=world)
    assert per_rank.tolist() == info["obs_per_rank"]                  # a point's observations never split


def test_plan_handles_giant_and_empty():
    # a point seen by 700 cameras (more than a chunk), a point with no observation, unused cameras
    n_cam, n_pt = 800, 6
    oc = np.concatenate([np.arange(700), [0, 1, 2], [5, 6]]).astype(np.int32)
    op = np.concatenate([np.zeros(700), [1, 1, 1], [3, 3]]).astype(np.int32)
    oxy = np.zeros((oc.size, 2))
    info = miba.plan_shards(n_cam, n_pt, oc, op, oxy, 2)
    assert info["n_valid"] == 705 and info["n_slots"] == 256 * 4
    owner = miba.plan_point_owner(n_cam, n_pt, oc, op, oxy, 2)
    assert owner[2] == -1 and owner[4] == -1 and owner[5] == -1 and owner[0] >= 0


def test_plan_rejects_bad_indices():
    oc = np.array([0, 7], dtype=np.int32)
    op = np.array([0, 0], dtype=np.int32)
    with pytest.raises(miba.MibaError):
        miba.plan_shards(3, 2, oc, op, np.zeros((2, 2)), 1)


_GLOO_WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from bundle_adjustment_amd import miba, synth
from oracle import oracle as O
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
pr = synth.make_problem(30, 2000, 9000, seed=6, shuffle_obs=True)
cams0, pts0 = synth.perturb(pr, 0.01, 0.1, seed=7)
owner = miba.plan_point_owner(30, 2000, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"], world)
mine = owner[pr["obs_pt"]] == rank
# every rank evaluates only the observations of the points it owns (cameras replicated) ...
ev = O.evaluate(cams0, pts0, pr["obs_cam"][mine], pr["obs_pt"][mine], pr["obs_xy"][mine])
col2 = np.zeros((30, 9))
np.add.at(col2, pr["obs_cam"][mine], (ev["Jc"] ** 2).sum(1))
buf = torch.from_numpy(np.concatenate([[ev["cost"]], ev["grad_cam"].ravel(), col2.ravel()]))
# ... and camera-sized quantities are completed by ONE all-reduce (what RCCL does on the GPUs)
dist.all_reduce(buf)
full = O.evaluate(cams0, pts0, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"])
col2_full = np.zeros((30, 9))
np.add.at(col2_full, pr["obs_cam"], (full["Jc"] ** 2).sum(1))
got = buf.numpy()
assert abs(got[0] - full["cost"]) <= 1e-12 * full["cost"]
assert np.abs(got[1:271] - full["grad_cam"].ravel()).max() <= 1e-12 * np.abs(full["grad_cam"]).max()
assert np.abs(got[271:] - col2_full.ravel()).max() <= 1e-12 * col2_full.max()
# point-sized quantities never leave their rank: the local point gradient is already complete
own_pts = owner == rank
assert np.abs(ev["grad_pt"][own_pts] - full["grad_pt"][own_pts]).max() <= 1e-12 * np.abs(full["grad_pt"]).max()
assert np.all(ev["grad_pt"][~own_pts] == 0.0)
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
"""


def test_point_sharding_world2_gloo(tmp_path):
    """world_size 2 on CPU (gloo): the sharding rule of miba_plan (points and their observations owned
    by one rank, cameras replicated, one all-reduce of camera-sized vectors) reproduces the
    single-process result."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "rank 0 ok" in out.stdout and "rank 1 ok" in out.stdout


def test_plan_threaded_layout_equals_serial():
    """Phase 2 of the tiler lays ranges of tiles out on worker threads (problems above 2^18 observations): the plan
    must not depend on the number of threads."""
    pr = synth.make_problem(300, 60000, 330000, seed=13, shuffle_obs=True)
    oc, op, oxy = pr["obs_cam"], pr["obs_pt"], pr["obs_xy"]
    res = []
    for thr in ("1", "2", "7"):
        os.environ["MIBA_PLAN_THREADS"] = thr
        try:
            info = miba.plan_shards(300, 60000, oc, op, oxy, 8)
            owner = miba.plan_point_owner(300, 60000, oc, op, oxy, 8)
        finally:
            os.environ.pop("MIBA_PLAN_THREADS", None)
        res.append((info, owner))
    assert res[0][0]["n_tiles"] >= 100
    for info, owner in res[1:]:
        assert info == res[0][0]
        assert np.array_equal(owner, res[0][1])


def test_multi_device_handle_needs_a_gpu():
    """miba_create_multi (one handle for several GPUs) is exported and, like every entry point, has no CPU fallback"""
    import ctypes as C
    lib = miba.lib()
    h = C.c_void_p()
    rc = lib.miba_create_multi(None, C.c_int32(2), C.c_int32(0), C.byref(h))
    if rc == 0:       # a GPU box: the handle works, then
        lib.miba_destroy(h)
        return
    assert rc == -2 and b"no HIP device" in lib.miba_last_error(None)
    assert lib.miba_create_multi(None, C.c_int32(9), C.c_int32(0), C.byref(h)) == -1   # 1..8 devices


def test_group_plan_is_the_rank_plan():
    """a handle for N devices builds ONE plan with world size N and hands rank r the tiles [shard_tile[r], shard_tile[r+1]):
    the same cut N one-rank-per-process handles make (both call build_plan with the same arguments): the shards partition
    the points, are balanced by observation count, and do not depend on which rank asks"""
    pr = synth.make_problem(60, 9000, 42000, seed=19, shuffle_obs=True)
    oc, op, oxy = pr["obs_cam"], pr["obs_pt"], pr["obs_xy"]
    for world in (2, 4, 8):
        info = miba.plan_shards(60, 9000, oc, op, oxy, world)
        owner = miba.plan_point_owner(60, 9000, oc, op, oxy, world)
        assert sum(info["obs_per_rank"]) == info["n_valid"] == 42000
        assert set(np.unique(owner)) == set(range(world))
        counts = np.bincount(owner[op], minlength=world)
        assert list(counts) == info["obs_per_rank"]
        assert max(info["obs_per_rank"]) <= 1.25 * min(info["obs_per_rank"]) + 4096


@pytest.mark.parametrize("chunks", ["2", "8", "31"])
def test_product_layout_is_a_permutation(chunks, monkeypatch):
    """The camera-major sub-tile layout of the product kernel (csrc/miba_plan.hpp build_cm_layout, used by csrc/miba_mf2.hpp):
    every observation of a non-giant point appears exactly once with its own (point, camera) pair, padding entries repeat a
    real pair, accumulator rows stay inside their camera's range, sub-tiles cover a tile's points in order.  Checked on the
    CPU by the library's own verifier (miba_plan_product_stats) for three tile sizes and a problem with giant tracks,
    filtered observations and a point seen twice by one camera."""
    monkeypatch.setenv("MIBA_TILE_CHUNKS", chunks)
    pr = synth.make_problem(90, 9000, 60000, seed=5)
    oc, op, xy = pr["obs_cam"].copy(), pr["obs_pt"].copy(), pr["obs_xy"].copy()
    # a giant track (more observations than a wavefront), filtered observations, a duplicate (point, camera) pair
    op[:200] = 17
    oc[:200] = np.arange(200) % 90
    op[300:320] = -1
    xy[330:340, 0] = np.nan
    oc[401] = oc[400]; op[401] = op[400]
    st = miba.plan_product_stats(90, 9000, oc, op, xy)
    assert st["ok"], st
    valid = (op >= 0) & ~np.isnan(xy[:, 0])
    counts = np.bincount(op[valid], minlength=9000)
    giant_obs = int(counts[counts > 64].sum())
    assert st["n_giant_obs"] == giant_obs and giant_obs >= 200
    assert st["n_obs"] == int(valid.sum()) - giant_obs
    assert st["n_entries"] >= st["n_obs"] and st["n_seg"] * 4 >= st["n_obs"]
    assert 1.0 <= st["cam_mult"] <= 8.0 and st["pt_mult"] >= 1.0


def _band_ring(n, w, ring):
    eu, ev = [], []
    for v in range(n):
        for d in range(1, w + 1):
            u = v + d
            if u < n:
                eu.append(v); ev.append(u)
            elif ring:
                eu.append(v); ev.append(u - n)
    return np.array(eu, np.int32), np.array(ev, np.int32)


def _symbolic_fill(n, eu, ev, pos):
    """independent restatement: eliminate in the given order, return the column patterns (as steps) and the fill count"""
    order = np.argsort(pos)
    adj = [set() for _ in range(n)]
    for u, v in zip(eu.tolist(), ev.tolist()):
        if u != v:
            adj[u].add(v); adj[v].add(u)
    done = np.zeros(n, bool)
    cols, fill = [], 0
    for v in order.tolist():
        nb = [u for u in adj[v] if not done[u]]
        cols.append(sorted(int(pos[u]) for u in nb))
        fill += len(nb)
        for a in nb:
            adj[a].update(b for b in nb if b != a)
        done[v] = True
    return cols, fill


@pytest.mark.parametrize("nested", [True, False])
@pytest.mark.parametrize("shape,n,w,arcs", [("ring", 400, 6, 3), ("band", 300, 4, 4), ("ring", 257, 9, 2), ("ring", 120, 5, 1), ("ring", 1500, 5, 12),
                                            ("band", 1200, 3, 32), ("ring", 2000, 6, 16)])
def test_tile_order_dissects_bands_and_rings(shape, n, w, arcs, nested):
    """the exact solve's elimination order (csrc/miba_plan.hpp, order_tile_graph; what Ceres' ordering does behind ceres::Solve at
    src/BundleAdjustment.cpp:108 for SPARSE_SCHUR): a permutation; the groups' steps are contiguous, the arcs (level 0) come before the
    separator sets, the sets level by level; a column of a group holds rows of its own group and of higher levels only (so the groups
    of one level can be factored side by side); the fill it reports is the fill of an independent symbolic elimination in that order,
    and close to the undissected order's.  nested: the cuts form a binary tree and every cut has a separator set of its own, whose
    level is the cut's height; else one set holds every separator row."""
    eu, ev = _band_ring(n, w, shape == "ring")
    r = miba.plan_tile_order(n, eu, ev, arcs, nested=nested)
    pos, st = r["pos"], r["arc_start"]
    gs, gl = r["group_start"], r["group_level"]
    assert sorted(pos.tolist()) == list(range(n))
    assert r["n_arcs"] == arcs and len(st) == arcs + 1 and st[0] == 0 and all(a <= b for a, b in zip(st, st[1:])) and st[-1] <= n
    assert gs[0] == 0 and gs[-1] == n and all(a <= b for a, b in zip(gs, gs[1:])) and len(gs) == len(gl) + 1
    assert gl[:arcs] == [0] * arcs and gs[:arcs] == st[:arcs] and all(a <= b for a, b in zip(gl, gl[1:]))
    if arcs > 1:
        assert gs[arcs] == st[-1] and all(l >= 1 for l in gl[arcs:])
        if not nested:
            assert len(gl) == arcs + 1
    cols, fill = _symbolic_fill(n, eu, ev, pos)
    assert fill == r["fill_tiles"]
    group_of_step = np.searchsorted(np.asarray(gs), np.arange(n), side="right") - 1
    for k, rows in enumerate(cols):
        g = group_of_step[k]
        for row in rows:
            assert row > k
            assert group_of_step[row] == g or gl[group_of_step[row]] > gl[g], (k, row)   # own group or a higher level, never a sibling
    if arcs > 1:
        assert 0 < n - st[-1] <= n // 3                             # separators exist and are small
        plain = miba.plan_tile_order(n, eu, ev, 1)
        # (an arc carries the separator rows of its low end along, as the plain order of a RING carries the rows that close it: about
        # the same fill on a ring, up to twice a plain band's)
        assert r["fill_tiles"] <= (1.3 if shape == "ring" else 2.0) * plain["fill_tiles"], (r["fill_tiles"], plain["fill_tiles"])
        # the model the symbolic phase chooses the number of arcs by (tile_order_cost): shorter chains of launches are cheaper
        assert r["cost_us"] < plain["cost_us"]
        steps = sum(max(b - a for a, b, l in zip(gs, gs[1:], gl) if l == lv) for lv in sorted(set(gl)))
        assert r["cost_us"] >= 38.0 * steps - 1e-6
        if nested and arcs >= 4:
            flat = miba.plan_tile_order(n, eu, ev, arcs, nested=False)
            assert steps < max(b - a for a, b in zip(st, st[1:-1] + [st[-1]])) + (n - st[-1]) and r["cost_us"] < flat["cost_us"]


def test_tile_order_without_locality_is_not_dissected():
    """a graph whose edges ignore the numbering has no small separator by index: one chain, plain minimum degree"""
    rng = np.random.default_rng(4)
    n = 200
    eu = rng.integers(0, n, 1200).astype(np.int32)
    ev = rng.integers(0, n, 1200).astype(np.int32)
    r = miba.plan_tile_order(n, eu, ev, 3)
    assert r["n_arcs"] == 1 and r["arc_start"] == [0, n]
    cols, fill = _symbolic_fill(n, eu, ev, r["pos"])
    assert fill == r["fill_tiles"]

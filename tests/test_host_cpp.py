"""The C++ host mirror of the reference's classes (include/bundleadjust, bundle_adjustment_amd/host):
BalDataloader and BAConstraint on the CPU, BundleAdjustment + app/main.cpp end to end on the GPU."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from bundle_adjustment_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = os.path.join(ROOT, "app", "ba_functor_check")
MAIN = os.path.join(ROOT, "app", "ba_main")
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    m = np.abs(b).max()
    return np.abs(a - b).max() / (m if m > 0 else 1.0)


def test_baconstraint_autodiff_matches_mpmath():
    """BAConstraint::create(...) -> AutoDiffCostFunction<BAConstraint,2,3,3,3,6> (the reference's
    construction, src/BAConstraint.cpp:82-86) evaluated with dual numbers, against the fixtures.
    Autodiff differentiates (1 - cos) literally, so tiny-but-not-small angles get 1e-9 here."""
    g = json.load(open(os.path.join(GOLD, "proj_golden.json")))
    lines = []
    for c in g["cases"]:
        lines.append(" ".join(c["cam_hex"] + c["X_hex"] + c["obs_hex"]))
    out = subprocess.run([CHECK], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    rows = [[float.fromhex(t) for t in ln.split()] for ln in out.stdout.strip().split("\n")]
    assert len(rows) == len(g["cases"])
    for row, c in zip(rows, g["cases"]):
        R = np.array([float(v) for v in c["r"]])
        JC = np.array([[float(v) for v in r_] for r_ in c["Jc"]])
        JP = np.array([[float(v) for v in r_] for r_ in c["Jp"]])
        obs = np.array([float.fromhex(h) for h in c["obs_hex"]])
        r = np.array(row[0:2]); jc = np.array(row[2:20]).reshape(2, 9); jp = np.array(row[20:26]).reshape(2, 3)
        theta = np.linalg.norm([float.fromhex(h) for h in c["cam_hex"][:3]])
        tol = 1e-10 if (c["small_angle"] or theta > 1e-3) else 1e-6
        assert np.abs(r - R).max() / max(1.0, np.abs(R + obs).max()) <= 1e-12
        for a, b in [(jc[:, 0:3], JC[:, 0:3]), (jc[:, 3:6], JC[:, 3:6]), (jc[:, 6:], JC[:, 6:]), (jp, JP)]:
            assert rel(a, b) <= tol


def test_bal_loader_parses_as_float_and_initializes(tmp_path, oracle):
    """Text -> float (src/BalDataloader.cpp:32-33,44-45,55-56) -> double; initialize() = the
    glibc-rand noise restated in the oracle (same libc here, seed 1 = unseeded)."""
    pr = synth.make_problem(5, 12, 34, seed=1)
    # un-rounded scalars in the file: the loader must round them through float
    rng = np.random.default_rng(0)
    raw = dict(pr)
    raw["cams"] = pr["cams"] + rng.normal(0, 1e-9, pr["cams"].shape)
    raw["pts"] = pr["pts"] + rng.normal(0, 1e-9, pr["pts"].shape)
    raw["obs_xy"] = pr["obs_xy"] + rng.normal(0, 1e-7, pr["obs_xy"].shape)
    path = str(tmp_path / "problem-5-12-pre.txt")
    synth.write_bal(path, raw)
    out = subprocess.run([CHECK, "--loader", path], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    tok = out.stdout.split("\n")
    assert tok[0].split() == ["5", "12", "34"]
    for i in range(34):
        c, p, x, y = tok[1 + i].split()
        assert (int(c), int(p)) == (pr["obs_cam"][i], pr["obs_pt"][i])
        assert float.fromhex(x) == float(np.float32(raw["obs_xy"][i, 0])) and float.fromhex(y) == float(np.float32(raw["obs_xy"][i, 1]))
    vals = np.array([float.fromhex(t) for t in tok[35:35 + 15 + 15 + 30 + 36]])
    R, T, K, X = oracle.bal_initialize(raw["cams"].astype(np.float32), raw["pts"].astype(np.float32), 0.01, 2.0, seed=1)
    assert np.array_equal(vals[0:15], R.ravel()) and np.array_equal(vals[15:30], T.ravel())
    assert np.array_equal(vals[30:60], K.ravel()) and np.array_equal(vals[60:96], X.ravel())
    assert tok[35 + 96].split() == ["255", "255", "255", "0", "0", "1"]


def test_bal_loader_throws_on_missing_file():
    out = subprocess.run([CHECK, "--loader", "/nonexistent/problem.txt"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 3 and "File could not be opened" in out.stdout


@pytest.mark.gpu
def test_app_main_end_to_end(tmp_path, oracle):
    """app/main.cpp: BalDataloader -> BundleAdjustment(ctor, createProblem, solve, WriteToPLYFile)
    on the GPU, final cost against the oracle started from the same (glibc-noise) state."""
    pr = synth.make_problem(20, 600, 2600, seed=5)
    path = str(tmp_path / "problem-20-600-pre.txt")
    synth.write_bal(path, pr)
    env = dict(os.environ, BA_SIGMA_CAM="0.005", BA_SIGMA_X="0.05", BA_ITERATION_PLY="1")
    out = subprocess.run([MAIN, path, "SPARSE_SCHUR", "10"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    m = re.search(r"FINAL_COST (\S+) ITERATIONS (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    final = float(m.group(1))
    R, T, K, X = oracle.bal_initialize(pr["cams"].astype(np.float32), pr["pts"].astype(np.float32), 0.005, 0.05, seed=1)
    cams0 = np.concatenate([R, T, K[:, 0:1], K[:, 4:5], K[:, 5:6]], 1)
    _, _, oits, osumm = oracle.solve(cams0, X, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                                     oracle.default_options(max_num_iterations=10, linear_solver=oracle.DENSE_SCHUR))
    assert abs(final - osumm["final_cost"]) <= 1e-6 * osumm["final_cost"], (final, osumm)
    # the reference writes one PLY per iteration (iteration -1 included) plus the final one
    assert os.path.exists(tmp_path / "mesh_SPARSE_SCHUR_-1.ply") and os.path.exists(tmp_path / "constrainTransReal.ply")
    head = open(tmp_path / "constrainTransReal.ply").read().split("\n")
    assert head[0] == "ply" and head[2] == "element vertex %d" % (600 + 20)
    assert head[10].endswith(" 0 255 0") and "Solver Summary" in out.stdout
    # one dump per iteration record (written by the background PLY thread, all on disk when solve() has returned),
    # every one complete; the dump of the last iteration holds the final state
    n_it = int(m.group(2))
    dumps = sorted(f for f in os.listdir(tmp_path) if f.startswith("mesh_SPARSE_SCHUR_"))
    assert len(dumps) >= n_it and all(len(open(tmp_path / f).read().split("\n")) == 10 + 620 + 1 for f in dumps)
    last = max(int(f[len("mesh_SPARSE_SCHUR_"):-4]) for f in dumps)
    assert open(tmp_path / ("mesh_SPARSE_SCHUR_%d.ply" % last)).read() == open(tmp_path / "constrainTransReal.ply").read()
    # a factorisation-based solver type on a small problem is served by the device Cholesky, and the report says so
    if os.environ.get("MIBA_DENSE", "1") != "0":
        assert "blocked Cholesky on the device" in out.stdout, out.stdout[-2000:]


@pytest.mark.gpu
def test_app_main_reference_constant_blocks(tmp_path, oracle):
    """BA_REFERENCE_CONSTANT_BLOCKS=1: createProblem marks what the reference marks (src/BundleAdjustment.cpp:77-79,
    88-90: rotation + intrinsics of every camera, all of camera 0); final cost against the oracle with the same mask."""
    pr = synth.make_problem(20, 600, 2600, seed=5)
    path = str(tmp_path / "problem-20-600-pre.txt")
    synth.write_bal(path, pr)
    env = dict(os.environ, BA_SIGMA_CAM="0.005", BA_SIGMA_X="0.05", BA_ITERATION_PLY="0", BA_REFERENCE_CONSTANT_BLOCKS="1")
    out = subprocess.run([MAIN, path, "SPARSE_SCHUR", "10"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    final = float(re.search(r"FINAL_COST (\S+) ITERATIONS (\d+)", out.stdout).group(1))
    R, T, K, X = oracle.bal_initialize(pr["cams"].astype(np.float32), pr["pts"].astype(np.float32), 0.005, 0.05, seed=1)
    cams0 = np.concatenate([R, T, K[:, 0:1], K[:, 4:5], K[:, 5:6]], 1)
    cc = np.zeros((20, 3), np.uint8)
    cc[:, 0] = 1; cc[:, 2] = 1; cc[0, :] = 1
    _, _, oits, osumm = oracle.solve(cams0, X, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                                     oracle.default_options(max_num_iterations=10, linear_solver=oracle.DENSE_SCHUR, cam_const=cc))
    assert abs(final - osumm["final_cost"]) <= 1e-6 * osumm["final_cost"], (final, osumm)


@pytest.mark.gpu
def test_app_main_with_loss_function(tmp_path, oracle):
    """BA_LOSS=huber:1.5: the LossFunction argument of AddResidualBlock, nullptr in the reference
    (src/BundleAdjustment.cpp:69), set to ceres::HuberLoss(1.5) for a file with gross outliers; final cost against
    the oracle with the same loss."""
    pr = synth.make_problem(20, 600, 2600, seed=5)
    oxy = pr["obs_xy"].copy()
    oxy[::8] += np.random.default_rng(2).normal(0, 35.0, oxy[::8].shape)
    pr = dict(pr, obs_xy=oxy.astype(np.float32).astype(np.float64))
    path = str(tmp_path / "problem-20-600-pre.txt")
    synth.write_bal(path, pr)
    env = dict(os.environ, BA_SIGMA_CAM="0.005", BA_SIGMA_X="0.05", BA_ITERATION_PLY="0", BA_LOSS="huber:1.5")
    out = subprocess.run([MAIN, path, "ITERATIVE_SCHUR", "10"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    final = float(re.search(r"FINAL_COST (\S+) ITERATIONS (\d+)", out.stdout).group(1))
    R, T, K, X = oracle.bal_initialize(pr["cams"].astype(np.float32), pr["pts"].astype(np.float32), 0.005, 0.05, seed=1)
    cams0 = np.concatenate([R, T, K[:, 0:1], K[:, 4:5], K[:, 5:6]], 1)
    _, _, oits, osumm = oracle.solve(cams0, X, pr["obs_cam"], pr["obs_pt"], pr["obs_xy"],
                                     oracle.default_options(max_num_iterations=10, linear_solver=oracle.ITERATIVE_SCHUR,
                                                            loss_type=oracle.LOSS_HUBER, loss_scale=1.5))
    assert abs(final - osumm["final_cost"]) <= 1e-6 * osumm["final_cost"], (final, osumm)


def test_bal_loader_threaded_parse_equals_serial(tmp_path):
    """The multi-threaded from_chars parser against the same parser on one thread, on a file large
    enough (> 1 MiB) to be cut into ranges; also '+'-signed numbers, CRLF line ends and a truncated file."""
    pr = synth.make_problem(30, 6000, 26000, seed=8)
    path = str(tmp_path / "problem-30-6000-pre.txt")
    synth.write_bal(path, pr)
    txt = open(path).read()
    assert len(txt) > (1 << 20)
    outs = []
    for thr in ("1", "7"):
        out = subprocess.run([CHECK, "--loader", path], capture_output=True, text=True, timeout=120,
                             env=dict(os.environ, BA_LOADER_THREADS=thr))
        assert out.returncode == 0, out.stderr
        outs.append(out.stdout)
    assert outs[0] == outs[1]
    tok = outs[0].split("\n")
    assert tok[0].split() == ["30", "6000", "26000"]
    c, p, x, y = tok[26000].split()   # last observation
    assert (int(c), int(p)) == (pr["obs_cam"][-1], pr["obs_pt"][-1])
    assert float.fromhex(x) == float(np.float32(pr["obs_xy"][-1, 0]))
    # '+' signs and CRLF
    lines = txt.split("\n")
    lines[1] = " ".join(("+" + t if not t.startswith("-") and i >= 2 else t) for i, t in enumerate(lines[1].split()))
    p2 = str(tmp_path / "crlf.txt")
    open(p2, "w", newline="").write("\r\n".join(lines))
    out = subprocess.run([CHECK, "--loader", p2], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout == outs[0]
    # truncated
    p3 = str(tmp_path / "trunc.txt")
    open(p3, "w").write(txt[: len(txt) // 2])
    out = subprocess.run([CHECK, "--loader", p3], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and "truncated" in out.stdout


def test_bal_loader_binary_cache(tmp_path):
    """BA_LOADER_CACHE=1: the second load comes from <file>.mibacache and is identical to the text
    parse (including a -1 point index); a changed text file invalidates the cache; without the knob nothing is written."""
    pr = synth.make_problem(12, 500, 2200, seed=9)
    path = str(tmp_path / "problem-12-500-pre.txt")
    synth.write_bal(path, pr)
    lines = open(path).read().split("\n")
    t = lines[5].split(); t[1] = "-1"; lines[5] = " ".join(t)    # an observation the solver will skip
    open(path, "w").write("\n".join(lines))
    plain = subprocess.run([CHECK, "--loader", path], capture_output=True, text=True, timeout=60)
    assert plain.returncode == 0 and not os.path.exists(path + ".mibacache")
    env = dict(os.environ, BA_LOADER_CACHE="1", BA_LOADER_DEBUG="1")
    first = subprocess.run([CHECK, "--loader", path], capture_output=True, text=True, timeout=60, env=env)
    assert first.returncode == 0 and "binary cache written" in first.stderr and os.path.exists(path + ".mibacache")
    second = subprocess.run([CHECK, "--loader", path], capture_output=True, text=True, timeout=60, env=env)
    assert second.returncode == 0 and "binary cache read" in second.stderr
    assert plain.stdout == first.stdout == second.stdout
    # stale cache: the text changes (size differs), the loader parses again and rewrites
    lines[7] = lines[7] + "0"
    open(path, "w").write("\n".join(lines))
    third = subprocess.run([CHECK, "--loader", path], capture_output=True, text=True, timeout=60, env=env)
    assert third.returncode == 0 and "binary cache read" not in third.stderr and "binary cache written" in third.stderr
    # a damaged cache is ignored
    open(path + ".mibacache", "r+b").truncate(100)
    fourth = subprocess.run([CHECK, "--loader", path], capture_output=True, text=True, timeout=60, env=env)
    assert fourth.returncode == 0 and "binary cache read" not in fourth.stderr and fourth.stdout == third.stdout


def test_ply_writer_text_equals_printf(tmp_path):
    """WriteToPLYFile (threaded, std::to_chars) writes exactly the text of the reference's
    `"%.4f %.4f %.4f r g b"` lines (src/BundleAdjustment.cpp:131-160): cameras' translations in green, then points;
    large enough (> 65536 lines) for the multi-threaded path, with values that round at the fourth decimal."""
    pr = synth.make_problem(6, 70000, 150000, seed=10)
    pr["pts"][:5] = [[0.00005, -0.00005, 1e9], [123456.78905, -0.99995, 0.5], [1e-9, -1e-9, 0.0], [2.5, -2.5, 1.00004999], [9.99995, 0.12345, -7.0]]
    path = str(tmp_path / "problem-6-70000-pre.txt")
    synth.write_bal(path, pr)
    ply = str(tmp_path / "out.ply")
    out = subprocess.run([CHECK, "--ply", path, ply], capture_output=True, text=True, timeout=120, cwd=str(tmp_path),
                         env=dict(os.environ, BA_SIGMA_CAM="0", BA_SIGMA_X="0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = out.stdout.split("\n")
    n_cam, n_pt = map(int, lines[0].split())
    vals = [[float.fromhex(t) for t in l.split()] for l in lines[1:1 + n_cam + n_pt]]
    got = open(ply).read().split("\n")
    assert got[:10] == ["ply", "format ascii 1.0", "element vertex %d" % (n_cam + n_pt), "property float x", "property float y",
                        "property float z", "property uchar red", "property uchar green", "property uchar blue", "end_header"]
    body = got[10:10 + n_cam + n_pt]
    assert len(body) == n_cam + n_pt and got[10 + n_cam + n_pt:] == [""]
    for i, (v, line) in enumerate(zip(vals, body)):
        color = "0 255 0" if i < n_cam else line.split(" ", 3)[3]
        assert line == "%.4f %.4f %.4f %s" % (v[0], v[1], v[2], color), (i, line, v)


@pytest.mark.gpu
def test_app_main_on_several_shards(tmp_path):
    """The reference's own call sequence (app/main.cpp:30-34) driving several shards through ONE BundleAdjustment object:
    BA_DEVICES=0,0,0 puts three ranks of a single-process group on GPU 0.  Same final cost as the one-shard run, per-iteration
    PLY dumps (state pulled from every shard inside the callback) complete and equal to the final file."""
    pr = synth.make_problem(24, 900, 4000, seed=15)
    path = str(tmp_path / "problem-24-900-pre.txt")
    synth.write_bal(path, pr)
    finals = {}
    for label, extra in (("one", {}), ("three", {"BA_DEVICES": "0,0,0"})):
        d = tmp_path / label
        d.mkdir()
        env = dict(os.environ, BA_SIGMA_CAM="0.005", BA_SIGMA_X="0.05", BA_ITERATION_PLY="1", **extra)
        out = subprocess.run([MAIN, path, "ITERATIVE_SCHUR", "8"], capture_output=True, text=True, timeout=120, cwd=str(d), env=env)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        m = re.search(r"FINAL_COST (\S+) ITERATIONS (\d+)", out.stdout)
        assert m, out.stdout[-2000:]
        finals[label] = (float(m.group(1)), int(m.group(2)))
        dumps = sorted(f for f in os.listdir(d) if f.startswith("mesh_ITERATIVE_SCHUR_"))
        assert all(len(open(d / f).read().split("\n")) == 10 + 924 + 1 for f in dumps)
        last = max(int(f[len("mesh_ITERATIVE_SCHUR_"):-4]) for f in dumps)
        assert open(d / ("mesh_ITERATIVE_SCHUR_%d.ply" % last)).read() == open(d / "constrainTransReal.ply").read()
    assert finals["one"][1] == finals["three"][1]
    assert abs(finals["one"][0] - finals["three"][0]) <= 1e-8 * finals["one"][0]


def test_reference_functor_autodiff_matches_mpmath():
    """The reference's checked-in functor (src/BAConstraint.cpp:15-80), as the host mirror evaluates it with dual numbers
    through BAConstraint::create, against the 80-digit vectors of the same text (camera model #2)."""
    g = json.load(open(os.path.join(GOLD, "proj2_golden.json")))
    lines = [" ".join(c["cam_hex"] + c["k4_hex"] + c["X_hex"] + c["obs_hex"]) for c in g["cases"]]
    out = subprocess.run([CHECK, "--model2"], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=60)
    assert out.returncode == 0
    rows = [[float.fromhex(t) for t in ln.split()] for ln in out.stdout.strip().split("\n")]
    assert len(rows) == len(g["cases"])
    for row, c in zip(rows, g["cases"]):
        R = np.array([float(v) for v in c["r"]])
        JC = np.array([[float(v) for v in r_] for r_ in c["Jc"]])
        JP = np.array([[float(v) for v in r_] for r_ in c["Jp"]])
        obs = np.array([float.fromhex(h) for h in c["obs_hex"]])
        r = np.array(row[0:2]); jr = np.array(row[2:8]).reshape(2, 3); jt = np.array(row[8:14]).reshape(2, 3)
        jp = np.array(row[14:20]).reshape(2, 3)
        theta = np.linalg.norm([float.fromhex(h) for h in c["cam_hex"][:3]])
        tol = 1e-10 if (c["small_angle"] or theta > 1e-3) else 1e-6   # autodiff differentiates (1 - cos) literally
        assert np.abs(r - R).max() / max(1.0, np.abs(obs - R).max()) <= 1e-12
        assert rel(jr, JC[:, 0:3]) <= tol and rel(jt, JC[:, 3:6]) <= tol and rel(jp, JP) <= tol
        assert row[20] == 0.0 and row[21] == 0.0   # the radial factor is computed and dropped: no dependence on k1, k2


@pytest.mark.gpu
def test_app_main_with_the_reference_functor_and_marks(tmp_path, oracle):
    """createProblem as it stands in the reference (src/BundleAdjustment.cpp:67-90): the checked-in functor
    (BA_CAMERA_MODEL=reference) and its constant-block marks (BA_REFERENCE_CONSTANT_BLOCKS=1: rotations and intrinsics of
    every camera, all of camera 0): only translations and points move.  A BAL file read through that functor is not a
    sensible scene (no sign flip, poses inverted), so the data here is generated for it: final cost against the oracle
    with the same model, marks and glibc noise."""
    rng = np.random.default_rng(21)
    n_cam, n_pt = 10, 300
    pr = synth.make_problem(n_cam, n_pt, 1400, seed=23)
    w = rng.normal(0, 0.2, (n_cam, 3)); t = rng.normal(0, 0.3, (n_cam, 3))
    f = rng.uniform(500, 560, n_cam)
    pts = np.stack([rng.uniform(-1.5, 1.5, n_pt), rng.uniform(-1.2, 1.2, n_pt), rng.uniform(3.0, 6.0, n_pt)], 1)
    cams = np.concatenate([w, t, f[:, None], np.zeros((n_cam, 2))], 1).astype(np.float32).astype(np.float64)
    pts = pts.astype(np.float32).astype(np.float64)
    k4 = np.stack([cams[:, 6], cams[:, 6], np.zeros(n_cam), np.zeros(n_cam)], 1)   # a BAL file carries no principal point
    xy = np.zeros((1400, 2))
    for i in range(1400):
        r, _, _ = oracle.project2(cams[pr["obs_cam"][i], 0:6], k4[pr["obs_cam"][i]], pts[pr["obs_pt"][i]], [0.0, 0.0], jac=False)
        xy[i] = -r
    xy = (xy + rng.normal(0, 0.3, xy.shape)).astype(np.float32).astype(np.float64)
    prob = dict(pr, cams=cams, pts=pts, obs_xy=xy)
    path = str(tmp_path / "problem-10-300-pre.txt")
    synth.write_bal(path, prob)
    env = dict(os.environ, BA_SIGMA_CAM="0.004", BA_SIGMA_X="0.03", BA_CAMERA_MODEL="reference", BA_REFERENCE_CONSTANT_BLOCKS="1")
    out = subprocess.run([MAIN, path, "SPARSE_SCHUR", "10"], capture_output=True, text=True, timeout=120, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    m = re.search(r"FINAL_COST (\S+) ITERATIONS (\d+)", out.stdout)
    assert m, out.stdout[-2000:]
    R, T, K, X = oracle.bal_initialize(cams.astype(np.float32), pts.astype(np.float32), 0.004, 0.03, seed=1)
    cams0 = np.concatenate([R, T, K[:, 0:1], K[:, 4:5], K[:, 5:6]], 1)
    cc = np.zeros((n_cam, 3), np.uint8); cc[:, 0] = 1; cc[:, 2] = 1; cc[0, :] = 1
    _, _, oits, osumm = oracle.solve(cams0, X, pr["obs_cam"], pr["obs_pt"], xy,
                                     oracle.default_options(max_num_iterations=10, linear_solver=oracle.DENSE_SCHUR, cam_const=cc, cam_k4=K[:, 0:4]))
    assert abs(float(m.group(1)) - osumm["final_cost"]) <= 1e-6 * osumm["final_cost"], (m.group(1), osumm)
    assert osumm["final_cost"] < 0.7 * osumm["initial_cost"]

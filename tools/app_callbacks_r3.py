"""Per-iteration callbacks of the C++ mirror at full size (app/ba_main on a BAL text file of a named configuration): solve() wall
time with the reference's per-iteration PLY dumps off, on with the callback between iterations (BA_ASYNC_CALLBACKS=0, what
ceres::Solve does) and on with the callback beside the solver (the default), and whether both write the same files.
usage: app_callbacks_r3.py [config] [iterations]"""
import hashlib, os, re, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
iters = sys.argv[2] if len(sys.argv) > 2 else "10"
n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
d = tempfile.mkdtemp()
path = os.path.join(d, "problem-%d-%d-pre.txt" % (n_cam, n_pt))
t = time.time()
synth.write_bal(path, pr)
print("wrote %.1f MB in %.1f s" % (os.path.getsize(path) / 1e6, time.time() - t), flush=True)
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'app', 'ba_main')


def run(tag, **env_over):
    wd = os.path.join(d, tag)
    os.makedirs(wd)
    env = dict(os.environ, BA_SIGMA_CAM="0.001", BA_SIGMA_X="0.05", BA_DEBUG_TIMING="1", **env_over)
    t0 = time.time()
    out = subprocess.run([exe, path, "ITERATIVE_SCHUR", iters], capture_output=True, text=True, cwd=wd, env=env)
    wall = time.time() - t0
    m = re.search(r"\[BundleAdjustment\] solve\s+([0-9.]+) s", out.stderr)
    tot = re.search(r"Total\s+([0-9.]+)", out.stdout)
    fc = re.search(r"FINAL_COST (\S+) ITERATIONS (\d+)", out.stdout)
    sums = {}
    for f in sorted(os.listdir(wd)):
        if f.endswith(".ply"):
            h = hashlib.md5()
            with open(os.path.join(wd, f), "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(blk)
            sums[f] = h.hexdigest()
    shutil.rmtree(wd, ignore_errors=True)
    print("%-28s rc %d  process wall %.2f s  solve() wall %s s  solver time (callback waits excluded) %s s  %s  %d ply files" % (
        tag, out.returncode, wall, m.group(1) if m else "?", tot.group(1) if tot else "?", fc.group(0) if fc else "?", len(sums)), flush=True)
    if out.returncode:
        print(out.stderr[-2000:])
    return sums


run("warm-up", BA_ITERATION_PLY="0")
run("ply_off", BA_ITERATION_PLY="0")
a = run("ply_on_between_iterations", BA_ITERATION_PLY="1", BA_ASYNC_CALLBACKS="0")
b = run("ply_on_beside_the_solver", BA_ITERATION_PLY="1")
print("same files byte for byte:", a == b and len(a) > 0, "(%d files)" % len(a))
shutil.rmtree(d, ignore_errors=True)

"""Write a synthetic BAL text file of a named configuration and time BalDataloader's parser on it
(app/ba_functor_check --loader-time) with several thread counts, next to the reference's serial ifstream loop."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bundle_adjustment_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "dubrovnik-356"
n_cam, n_pt, n_obs, seed = synth.CONFIGS[cfg]
pr = synth.make_problem(n_cam, n_pt, n_obs, seed=seed)
path = os.path.join(tempfile.gettempdir(), "bal_%s.txt" % cfg)
t = time.time()
synth.write_bal(path, pr)
print("wrote %s: %.1f MB in %.1f s" % (path, os.path.getsize(path) / 1e6, time.time() - t), flush=True)
exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'app', 'ba_functor_check')
for thr in ("1", "4", "16"):
    r = subprocess.run([exe, "--loader-time", path], capture_output=True, text=True, env=dict(os.environ, BA_LOADER_THREADS=thr))
    print("threads", thr, r.stdout.strip(), flush=True)
for label in ("cache write", "cache read"):
    r = subprocess.run([exe, "--loader-time", path], capture_output=True, text=True, env=dict(os.environ, BA_LOADER_CACHE="1"))
    print(label, r.stdout.strip(), flush=True)
os.remove(path)
os.remove(path + ".mibacache")

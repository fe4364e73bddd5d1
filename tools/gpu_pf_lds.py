"""PF kernel comparison on the GPU: LDS-resident kernel vs device-memory-table kernel (timing + agreement)."""
import sys, os, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

def run(mode, W, n):
    code = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
from scanfold_amd import _lib
W, n = %d, %d
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (n, W))]
eng = _lib.Engine(0)
eng.pf_batch(arr[:512])
t0 = time.time(); r = eng.pf_batch(arr); t1 = time.time()
print("TIME %%.4f" %% (t1 - t0))
np.savez("/tmp/pf_%%s_%%d.npz" %% (%r, W), dG=r["dG"], mbd=r["mean_bp_dist"], cd=r["centroid_dist"], cen=np.array(r["centroid"]))
''' % (ROOT, W, n, mode)
    env = dict(os.environ, SCANFOLD_PF_KERNEL=mode)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    if out.returncode:
        print(out.stdout, out.stderr)
        raise SystemExit(1)
    return float(out.stdout.split("TIME")[1])

for W, n in ((120, 8192), (100, 4096), (60, 4096), (31, 2048)):
    tl = run("lds", W, n)
    tg = run("global", W, n)
    a = np.load("/tmp/pf_lds_%d.npz" % W); b = np.load("/tmp/pf_global_%d.npz" % W)
    err = max(np.abs(a["dG"] - b["dG"]).max(), np.abs(a["mbd"] - b["mbd"]).max(), np.abs(a["cd"] - b["cd"]).max())
    print("W=%d n=%d: lds %.3fs  global %.3fs  max|diff| %.2e  centroids equal %s" % (W, n, tl, tg, err, bool((a["cen"] == b["cen"]).all())), flush=True)

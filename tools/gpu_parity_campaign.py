"""Large randomized parity campaign on the GPU box: HIP kernels (every MFE kernel mode, PF LDS + device-table kernels)
against the oracle, many widths / compositions.  Prints one line per case; exits non-zero on any mismatch."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib, params
from oracle import oracle
oracle.build(); oracle.set_params(params.default_params())
eng = _lib.get_engine(0)
rng = np.random.default_rng(20261003)
bad = 0
ALPH = np.frombuffer(b"ACGU", dtype=np.uint8)
def seqs(n, W, p):
    return ALPH[rng.choice(4, size=(n, W), p=p)]
comps = {"uniform": [.25, .25, .25, .25], "GC-rich": [.1, .4, .4, .1], "AU-rich": [.4, .1, .1, .4], "GU-rich": [.05, .05, .45, .45],
         "GC-only": [0, .5, .5, 0]}  # GC-only: half of all cells pair (more than 64 pairable cells on a split step: the merged helper's second chunk)
t0 = time.time()
for W, n in ((120, 40000), (100, 8000), (80, 4000), (64, 8000), (37, 8000), (127, 4000), (128, 4000), (146, 1500), (160, 1500), (200, 1500),
             (16, 4000)):
    for cname, p in comps.items():
        arr = seqs(n if cname == "uniform" else n // 4, W, p)
        ref = oracle.mfe_batch(arr)
        for mode in (0,):
            eng.set_kernel_mode(mode)
            e = eng.mfe_batch(arr)
            nb = int((e != ref).sum())
            bad += nb
            print("MFE W=%d %s n=%d mode=%d mismatches=%d min=%d" % (W, cname, len(arr), mode, nb, int(ref.min())), flush=True)
        eng.set_kernel_mode(0)
# structures (traceback) + partition function
for W, n in ((120, 1500), (90, 600), (61, 600), (30, 600)):
    arr = seqs(n, W, comps["uniform"])
    e, db = eng.mfe_trace_batch(arr)
    r = eng.pf_batch(arr)
    nb = 0; worst = 0.0
    for k in range(n):
        s = bytes(arr[k]).decode()
        odb, oe = oracle.mfe(s)
        o = oracle.pf(s)
        if (db[k], e[k]) != (odb, oe) or o["centroid"] != r["centroid"][k]: nb += 1
        worst = max(worst, abs(o["dG"] - r["dG"][k]), abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]), abs(o["centroid_dist"] - r["centroid_dist"][k]))
    bad += nb + (worst > 1e-8)
    print("TRACE+PF W=%d n=%d string mismatches=%d max|dPF|=%.2e" % (W, n, nb, worst), flush=True)
print("total mismatches", bad, "elapsed %.0fs" % (time.time() - t0))
sys.exit(1 if bad else 0)

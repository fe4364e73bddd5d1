"""Shared inside tables of sf_scan (sf_pf_lds.hip.h, SH) against stand-alone folds of the same windows:
several widths, steps and base compositions; every window vs the same kernel on independent rows, a sample vs the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib, params
from oracle import oracle
oracle.set_params(params.default_params())
eng = _lib.get_engine(0)
rng = np.random.default_rng(2024)
total = 0
t0 = time.time()
for W, step, L, comp in ((120, 1, 2600, "ACGU"), (120, 1, 1500, "GGCCAU"), (120, 3, 3000, "ACGU"), (120, 10, 6000, "AAUUGC"),
                         (120, 30, 9000, "ACGU"), (90, 1, 1800, "ACGUN"), (61, 2, 1500, "ACGU"), (33, 1, 900, "GGUUAC"), (16, 1, 700, "ACGU")):
    tr = "".join(comp[k] for k in rng.integers(0, len(comp), L))
    nwin = (L - W) // step + 1
    res = eng.scan(tr, W, step, 0, nwin, 1, 1, 5)
    wins = [tr[w * step:w * step + W] for w in range(nwin)]
    alone = eng.pf_batch(wins)
    bad = sum(1 for w in range(nwin) if res["centroid"][w] != alone["centroid"][w])
    dev = max(float(np.max(np.abs(res["ens_div"] - alone["mean_bp_dist"]))), float(np.max(np.abs(res["ens_dG"] - alone["dG"]))))
    obad = 0
    for w in range(0, nwin, 25):
        o = oracle.pf(wins[w])
        if o["centroid"] != res["centroid"][w] or abs(o["mean_bp_dist"] - res["ens_div"][w]) > 1e-8 or abs(o["dG"] - res["ens_dG"][w]) > 1e-8:
            obad += 1
    total += bad + obad + (dev > 1e-10)
    print("W=%d step=%d windows=%d: centroid mismatches vs stand-alone %d, max |delta| %.2e, oracle mismatches (sample) %d" % (W, step, nwin, bad, dev, obad), flush=True)
print("total mismatches %d  (%.0f s)" % (total, time.time() - t0))

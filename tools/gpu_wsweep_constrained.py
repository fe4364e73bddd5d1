"""Constrained folds (sf_fold_constrained) at every window width 16..256: the LDS kernels with the constraint at their pair-type
seam (kernel mode 0; W <= 250 MFE, W <= 120 partition function) against the general kernels (mode 1), and both against the
oracle on a sample.  Hard constraints with pairable bracket pairs, x < > |; Deigan pseudo-energies on odd widths."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scanfold_amd import _lib, params
from oracle import oracle
from test_constraints import canonical_constraint, random_constraint, rseq
oracle.set_params(params.default_params())
eng = _lib.get_engine(0)
rng = np.random.default_rng(424242)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
total = 0
t0 = time.time()
for W in range(16, 257, step):
    n = 48
    seqs = [rseq(rng, W) for _ in range(n)]
    cons = [canonical_constraint(rng, s, 4) for s in seqs] if W % 5 else [random_constraint(rng, W, 3) for _ in range(n)]
    sc = rng.integers(-60, 40, (n, W)).astype(np.int32) if W % 2 else None
    eng.set_kernel_mode(1)
    r1 = eng.fold_constrained(seqs, cons, sc)
    eng.set_kernel_mode(0)
    r0 = eng.fold_constrained(seqs, cons, sc)
    bad = sum(1 for k in range(n) if r0["structure"][k] != r1["structure"][k] or r0["centroid"][k] != r1["centroid"][k]) \
        + int((r0["mfe"] != r1["mfe"]).sum()) + int((np.abs(np.asarray(r0["dG"]) - np.asarray(r1["dG"])) > 1e-9).sum()) \
        + int((np.abs(np.asarray(r0["mean_bp_dist"]) - np.asarray(r1["mean_bp_dist"])) > 1e-9).sum())
    obad = 0
    for k in range(0, n, 12):
        oracle.set_constraint(cons[k], None if sc is None else sc[k])
        if oracle.mfe(seqs[k]) != (r0["structure"][k], int(r0["mfe"][k])): obad += 1
        oracle.set_constraint(cons[k], None)
        o = oracle.pf(seqs[k])
        if o["centroid"] != r0["centroid"][k] or abs(o["mean_bp_dist"] - r0["mean_bp_dist"][k]) > 1e-8: obad += 1
    oracle.set_constraint(None, None)
    total += bad + obad
    if bad or obad or W % 16 == 0:
        print("W=%d  LDS vs general kernels: %d differences   vs oracle (4 folds): %d" % (W, bad, obad), flush=True)
print("total mismatches %d  (%.0f s)" % (total, time.time() - t0))
sys.exit(1 if total else 0)

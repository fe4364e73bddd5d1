"""Every window width 16..256: MFE energies (three kernel passes over the same batch, so one persistent workgroup
folds several sequences in a row), traceback strings and partition-function outputs against the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib, params
from oracle import oracle
oracle.set_params(params.default_params())
eng = _lib.get_engine(0)
rng = np.random.default_rng(12345)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 1
total = 0
t0 = time.time()
for W in range(16, 257, step):
    n = 1500 if W <= 130 else 600   # > 1024 workgroups' worth: some workgroups fold two sequences
    comp = [b"ACGU", b"GGCCAU", b"AAUUGC", b"GGUUAC"][W % 4]
    arr = np.frombuffer(comp, dtype=np.uint8)[rng.integers(0, len(comp), (n, W))]
    ref = oracle.mfe_batch(arr)
    bad = 0
    for rep in range(2):
        bad += int((eng.mfe_batch(arr) != ref).sum())
    nt = 24
    e, db = eng.mfe_trace_batch(arr[:nt])
    rdb = [oracle.mfe(bytes(arr[k]).decode())[0] for k in range(nt)]
    sbad = sum(1 for k in range(nt) if db[k] != rdb[k]) + int((e != ref[:nt]).sum())
    pf = eng.pf_batch(arr[:nt])
    pbad = 0
    for k in range(nt):
        o = oracle.pf(bytes(arr[k]).decode())
        if (o["centroid"] != pf["centroid"][k] or abs(o["mean_bp_dist"] - pf["mean_bp_dist"][k]) > 1e-9
                or abs(o["dG"] - pf["dG"][k]) > 1e-9):
            pbad += 1
    total += bad + sbad + pbad
    if bad or sbad or pbad or W % 16 == 0:
        print("W=%d energies bad=%d  trace bad=%d  pf bad=%d" % (W, bad, sbad, pbad), flush=True)
print("total mismatches %d  (%.0f s)" % (total, time.time() - t0))

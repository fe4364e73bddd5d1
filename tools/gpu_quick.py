"""Ad-hoc GPU check used during development: parity vs oracle + rough timings."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scanfold_amd import _lib, params
from oracle import oracle

p = params.default_params()
oracle.set_params(p)
eng = _lib.get_engine(0)
print(eng.device_name(), flush=True)
rng = np.random.default_rng(0)
for W, n in ((30, 64), (120, 512), (200, 64)):
    arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (n, W))]
    t0 = time.time(); e = eng.mfe_batch(arr); t1 = time.time()
    ref = oracle.mfe_batch(arr)
    print("W", W, "n", n, "parity", bool((e == ref).all()), "ndiff", int((e != ref).sum()), "t %.3fs" % (t1 - t0), flush=True)
    m = min(n, 64)
    e2, db = eng.mfe_trace_batch(arr[:m])
    odb = [oracle.mfe(bytes(r).decode())[0] for r in arr[:m]]
    print("  trace parity", db == odb, bool((e2 == ref[:m]).all()), flush=True)
    t0 = time.time(); r = eng.pf_batch(arr[:m]); t1 = time.time()
    ok = True
    for k in range(m):
        o = oracle.pf(bytes(arr[k]).decode())
        ok &= abs(o['dG'] - r['dG'][k]) < 1e-8 and o['centroid'] == r['centroid'][k] and abs(o['mean_bp_dist'] - r['mean_bp_dist'][k]) < 1e-8
    print("  pf parity", ok, "t %.3fs" % (t1 - t0), flush=True)
for n in (4096, 32768, 262144):
    arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (n, 120))]
    eng.mfe_batch(arr[:256])
    eng.prof_reset()
    t0 = time.time(); e = eng.mfe_batch(arr); t1 = time.time()
    ms, nl, nf = eng.prof_get()
    print("mfe n", n, "wall %.3fs kernel %.1f ms -> %.0f folds/s" % (t1 - t0, ms, nf / (ms / 1e3)), flush=True)
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (2048, 120))]
t0 = time.time(); r = eng.pf_batch(arr); t1 = time.time()
print("pf 2048 x120: %.3fs" % (t1 - t0))
t0 = time.time(); r = eng.mfe_trace_batch(arr); t1 = time.time()
print("trace 2048 x120: %.3fs" % (t1 - t0))
import __graft_entry__
__graft_entry__.smoke()

"""`scan.py -c` against the plain scan on cfg3 (30 kb, W=120, step=1, r=100): the constrained native folds run on the LDS
kernels (round 3) instead of the general device-memory kernels (round 2: about +1 s per 30 k windows)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib, scan as scanmod
eng = _lib.Engine(0)
seq = "".join("ACGU"[k] for k in np.random.default_rng(3).integers(0, 4, 30000))
rng = np.random.default_rng(1)
cons = "".join(rng.choice(list(".....x<>|"), len(seq)))
react = None
def med(f, n=3):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
t_plain = med(lambda: scanmod.scan_record(seq, 120, 1, 100, "di", 37, eng, seed=1))
t_cons = med(lambda: scanmod.scan_record(seq, 120, 1, 100, "di", 37, eng, seed=1, constraints=cons))
eng.set_kernel_mode(1)
t_cons_general = med(lambda: scanmod.scan_record(seq, 120, 1, 100, "di", 37, eng, seed=1, constraints=cons), 1)
print("cfg3 scan_record: plain %.3f s; with -c (x < > | constraint line) %.3f s = %.2f x plain; kernel mode 1 (general kernels, "
      "every fold) %.3f s" % (t_plain, t_cons, t_cons / t_plain, t_cons_general))

"""`scan.py -t 25` against the plain scan on cfg3 (30 kb, W=120, step=1, r=100) with a parameter set that has enthalpies (the
shipped table re-emitted with synthetic enthalpy sections, tests/par_util.py): the native windows fold at 25 C, the shuffles at
37 C, chunk after chunk — the library keeps both models resident, so the alternation costs pointer switches, not reloads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scanfold_amd import _lib, params, scan as scanmod
from par_util import par_text, synthetic_enthalpies
base = params.default_params()
p = params.parse_par_text(par_text(base.rec, synthetic_enthalpies(base.rec, 9)), source="synthetic.par")
eng = _lib.Engine(0, paramset=p)
seq = "".join("ACGU"[k] for k in np.random.default_rng(3).integers(0, 4, 30000))
def med(f, n=3):
    f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
t37 = med(lambda: scanmod.scan_record(seq, 120, 1, 100, "di", 37, eng, seed=1))
t25 = med(lambda: scanmod.scan_record(seq, 120, 1, 100, "di", 25, eng, seed=1))
t0 = time.perf_counter()
for _ in range(20):
    eng.set_temperature(25); eng.set_temperature(37)
sw = (time.perf_counter() - t0) / 40
print("cfg3 scan_record: -t 37 %.3f s; -t 25 %.3f s = %.3f x; one switch between the two resident models %.3f ms" % (
    t37, t25, t25 / t37, sw * 1e3))

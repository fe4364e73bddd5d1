"""MFE kernel timing per kernel mode (0 default, 2 packed cells, 3 two folds per workgroup) + agreement."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
modes = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 2, 3]
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (n, W))]
eng = _lib.get_engine(0)
ref = None
for mode in modes:
    eng.set_kernel_mode(mode)
    eng.mfe_batch(arr[:2048])
    eng.prof_reset()
    e = eng.mfe_batch(arr)
    ms, nl, nf = eng.prof_get()
    if ref is None: ref = e
    print("mode %d W %d n %d kernel %.1f ms -> %.0f folds/s  same=%s" % (mode, W, n, ms, nf / ms * 1e3, bool((e == ref).all())), flush=True)
eng.set_kernel_mode(0)

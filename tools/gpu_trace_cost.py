"""Cost of the in-kernel traceback: n folds of W nt with and without structures (HIP-event time of the MFE kernel)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 29881
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
eng = _lib.Engine(0)
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (n, W))]
eng.mfe_batch(arr[:2048]); eng.mfe_trace_batch(arr[:2048])
for name, fn in (("energies only", eng.mfe_batch), ("with traceback", eng.mfe_trace_batch)):
    for rep in range(2):
        eng.prof_reset()
        fn(arr)
        ms, nl, nf = eng.prof_get()
    print("%-15s W %d n %d kernel %.2f ms -> %.0f folds/s (%.1f us per fold and workgroup slot)" % (name, W, n, ms, nf / ms * 1e3, ms * 1e3 * 1024 / nf))

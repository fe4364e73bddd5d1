"""Runs only the batched MFE kernel on n folds of W nt (for counter collection).  SCANFOLD_LIB=<path> picks a build variant."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
path = os.environ.get("SCANFOLD_LIB")
if path:
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=os.path.join(ROOT, path))
else:
    eng = _lib.Engine(0)
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (n, W))]
eng.mfe_batch(arr[:1024])
eng.prof_reset()
eng.mfe_batch(arr)
ms, nl, nf = eng.prof_get()
print("%s W %d n %d kernel %.1f ms -> %.0f folds/s" % (path or "product", W, n, ms, nf / ms * 1e3))

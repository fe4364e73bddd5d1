"""Run the PF kernel alone (for rocprofv3 counter passes)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (n, W))]
eng = _lib.Engine(0)
eng.pf_batch(arr[:256])
t0 = time.time(); eng.pf_batch(arr); t1 = time.time()
print("pf %d x %d: %.4fs" % (n, W, t1 - t0))

"""Thread scaling of the CPU twin on the box's host cores (no GPU, no torch): windows/s at several thread counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle
from scanfold_amd import params
oracle.build(); oracle.set_params(params.default_params())
seq = "".join("ACGU"[k] for k in np.random.default_rng(3).integers(0, 4, 30000))
W, r = 120, 100
for nt in [int(x) for x in (sys.argv[1:] or ["1", "8", "32", "64", "128"])]:
    n = max(nt * 2, 8)
    rows = np.frombuffer(b"NACGU", dtype=np.uint8)[oracle.shuffle_windows(seq, W, 1, 0, n, r, 1, 5)]
    oracle.twin_scan_windows(rows[: (r + 1) * min(n, nt)], min(n, nt), r, nthreads=nt)  # warm the workspaces
    t0 = time.perf_counter(); oracle.twin_scan_windows(rows, n, r, nthreads=nt); t = time.perf_counter() - t0
    print("threads %4d  windows %4d  %.2f s  -> %.1f windows/s  (%.3f s per window per thread)  OMP_PROC_BIND=%s" %
          (nt, n, t, n / t, t * nt / n, os.environ.get("OMP_PROC_BIND")), flush=True)

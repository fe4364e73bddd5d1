"""A/B of the device-table partition-function kernel (sf_pf_fast_kernel, 120 < W <= 256): product library against every
tools/abl_*.so — wall time of pf_batch on n random W-mers and agreement of the outputs with the first library's."""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
W = int(sys.argv[2]) if len(sys.argv) > 2 else 200
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (n, W))]
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so")))
ref = None
for lib in libs:
    if lib:
        _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=lib) if lib else _lib.Engine(0)
    eng.pf_batch(arr[:512])
    best = 1e9
    for rep in range(2):
        t0 = time.time(); o = eng.pf_batch(arr); best = min(best, time.time() - t0)
    if ref is None:
        ref = o
    ddg = float(np.abs(o["dG"] - ref["dG"]).max()); dmb = float(np.abs(o["mean_bp_dist"] - ref["mean_bp_dist"]).max())
    same = o["centroid"] == ref["centroid"]
    print("%-24s W %d n %d pf_batch %.1f ms -> %.0f folds/s   max|d dG| %.2e  max|d ED| %.2e  centroids equal %s" %
          (os.path.basename(lib or "libscanfold_hip.so"), W, n, best * 1e3, n / best, ddg, dmb, same), flush=True)

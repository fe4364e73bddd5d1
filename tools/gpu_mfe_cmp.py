"""MFE kernel timing for the product library and every tools/abl_*.so build variant."""
import sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
W = int(sys.argv[2]) if len(sys.argv) > 2 else 120
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (n, W))]
ref = None
for path in [_lib.LIB_PATH] + sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so"))):
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=path)
    eng.mfe_batch(arr[:1024])
    eng.prof_reset()
    e = eng.mfe_batch(arr)
    ms, nl, nf = eng.prof_get()
    if ref is None: ref = e
    print("%-24s W %d n %d kernel %.1f ms -> %.0f folds/s  same=%s" % (os.path.basename(path), W, n, ms, nf / ms * 1e3, bool((e == ref).all())), flush=True)
    eng.shutdown()

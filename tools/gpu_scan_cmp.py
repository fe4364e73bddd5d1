"""The cfg3 scan (30 kb synthetic transcript, W=120, step 1, 100 di-shuffles) for the product library and every
tools/abl_*.so build variant: HIP-event time of the MFE kernel launches and an energy checksum."""
import sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
W = int(sys.argv[1]) if len(sys.argv) > 1 else 120
flags = int(os.environ.get("SCAN_FLAGS", "0"))  # 1: no partition function, 2: no traceback
r = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seq = "".join("ACGU"[k] for k in np.random.default_rng(7).integers(0, 4, 30000))
n = len(seq) - W + 1
for path in [_lib.LIB_PATH] + sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so"))):
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=path)
    eng.scan(seq, W, 1, 0, 512, 10, _lib.SHUFFLE_DI, 1, raw=True)
    for rep in range(2):
        eng.prof_reset()
        res = eng.scan(seq, W, 1, 0, n, r, _lib.SHUFFLE_DI, 2026, flags, raw=True)
        ms, nl, nf = eng.prof_get()
    print("%-24s W %d r %d: %d MFE folds in %d launches, %.1f ms -> %.0f folds/s  checksum %d" % (
        os.path.basename(path), W, r, nf, nl, ms, nf / ms * 1e3, int(res["energies"].sum(dtype=np.int64))), flush=True)
    eng.shutdown()

import sys, os, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (n, 120))]
libs = [("base", _lib.LIB_PATH)] + [(os.path.basename(p)[4:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so")))]
for name, path in libs:
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=path)
    eng.mfe_batch(arr[:2048])
    eng.prof_reset()
    eng.mfe_batch(arr)
    ms, nl, nf = eng.prof_get()
    print("%-8s %8.1f ms  %9.0f folds/s" % (name, ms, nf / ms * 1e3), flush=True)
    eng.shutdown()

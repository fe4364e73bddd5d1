import sys, os, time, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (8192, 120))]
for path in [_lib.LIB_PATH] + sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so"))):
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=path)
    eng.pf_batch(arr[:512])
    t0 = time.time(); eng.pf_batch(arr); t1 = time.time()
    print(os.path.basename(path), "pf 8192: %.3fs" % (t1 - t0), flush=True)
    eng.shutdown()

"""Phase cycle counts of the LDS partition-function kernel from an -DSF_STAMP build (tools/stamp.so)."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
_lib._share_hip_runtime_with_torch()
eng = _lib.Engine(0, lib_path=os.path.join(ROOT, "tools", "stamp.so"))
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (8192, 120))]
eng.pf_batch(arr)
out = (ctypes.c_ulonglong * 200)()
eng.lib.sf_debug_stamps.argtypes = [ctypes.c_void_p]
eng.lib.sf_debug_stamps(out)
a = np.array(list(out)[192:200], dtype=np.float64)
print("per fold (block 0), s_memtime ticks: inside %.0f | exterior %.0f | outside %.0f  (%d folds)" % (a[0] / a[3], a[1] / a[3], a[2] / a[3], a[3]))

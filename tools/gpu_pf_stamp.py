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
out = (ctypes.c_ulonglong * 224)()
eng.lib.sf_debug_stamps.argtypes = [ctypes.c_void_p]
eng.lib.sf_debug_stamps(out)
a = np.array(list(out)[192:224], dtype=np.float64)
print("per fold (block 0), s_memtime ticks: inside %.0f | exterior %.0f | outside %.0f  (%d folds)" % (a[0] / a[3], a[1] / a[3], a[2] / a[3], a[3]))
print("work before the first barrier of a column, per fold: inside teams 0-3: %s | outside teams 0-3: %s" % (
    " ".join("%.0f" % (x / a[3]) for x in a[8:12]), " ".join("%.0f" % (x / a[3]) for x in a[16:20])))
print("team 2 between the two barriers, per fold: inside %.0f | outside %.0f" % (a[20] / a[3], a[21] / a[3]))

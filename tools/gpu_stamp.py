import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
_lib._share_hip_runtime_with_torch()
eng = _lib.Engine(0, lib_path=os.path.join(ROOT, "tools", "stamp.so"))
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(0).integers(0, 4, (65536, 120))]
eng.mfe_batch(arr)
out = (ctypes.c_ulonglong * 224)()
eng.lib.sf_debug_stamps.argtypes = [ctypes.c_void_p]
eng.lib.sf_debug_stamps(out)
a = np.array(list(out)[:64], dtype=np.float64).reshape(8, 8)
st = np.array(list(out)[64:], dtype=np.float64)
nfold = 65536 / 1024
print("per fold (block 0), s_memtime ticks: cell | barrier1 | oddfinal | pre-exchange(d0>=58) | exterior+trace | steps | cell(d0>=58) | steps(d0>=58)")
print('exterior (wave 0): sweep | last columns + result | table fill + barriers before the sweep')
print(' '.join('%9.0f' % x for x in a[4][:3] / nfold))
for w in range(4):
    r = a[w] / nfold
    print("wave", w, " ".join("%9.0f" % x for x in r))
print("per step, wave 0 (d0: ticks whole step | wave 1 work before the exchange barrier):")
print(" ".join("%d:%d|%d" % (2 * k, st[k] / nfold, st[64 + k] / nfold) for k in range(2, 60)))

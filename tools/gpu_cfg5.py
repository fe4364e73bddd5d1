"""BASELINE config 5 on ONE GPU: 30 kb transcript, W=200, step=1, r=1000 di-shuffles, partition function.
usage: gpu_cfg5.py [n_windows (default: all 29801)]   -> one JSON line (whole step, the step without PF, PF alone)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
eng = _lib.Engine(0)
seq = "".join("ACGU"[k] for k in np.random.default_rng(3).integers(0, 4, 30000))
W, r = 200, 1000
n_all = len(seq) - W + 1
n = int(sys.argv[1]) if len(sys.argv) > 1 else n_all
eng.scan(seq, W, 1, 0, 64, 10, _lib.SHUFFLE_DI, 1)  # warm-up
eng.prof_reset()
t0 = time.perf_counter()
res = eng.scan(seq, W, 1, 0, n, r, _lib.SHUFFLE_DI, 2026, raw=True)
t_all = time.perf_counter() - t0
ms, nl, nf = eng.prof_get()
eng.prof_stop()
t0 = time.perf_counter()
eng.scan(seq, W, 1, 0, n, 0, _lib.SHUFFLE_DI, 2026, _lib.SCAN_NO_TRACE, raw=True)  # native MFE + PF only
t_pf = time.perf_counter() - t0
t0 = time.perf_counter()
eng.scan(seq, W, 1, 0, n, 0, _lib.SHUFFLE_DI, 2026, _lib.SCAN_NO_TRACE | _lib.SCAN_NO_PF, raw=True)
t_nat = time.perf_counter() - t0
print(json.dumps(dict(config="cfg5: 30 kb, W=200, step=1, r=1000, di, PF; 1 GPU", windows=n, of=n_all, step_s=t_all,
                      windows_per_s=n / t_all, mfe_kernel_ms=ms, mfe_launches=nl, mfe_folds=nf,
                      mfe_folds_per_s=nf / ms * 1e3, pf_s=t_pf - t_nat, pf_share_of_step=(t_pf - t_nat) / t_all,
                      checksum=int(res["energies"].sum(dtype=np.int64)), device=eng.device_name())))

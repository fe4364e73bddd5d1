"""Partition-function time inside a cfg3-shaped scan (30 kb synthetic transcript, W=120, step 1: consecutive windows share
their inside tables) for the product library and every tools/abl_*.so build variant: wall time of a scan with one shuffle
with and without the partition function, their difference, and checksums of the PF outputs (must be identical)."""
import sys, os, glob, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib
W = int(sys.argv[1]) if len(sys.argv) > 1 else 120
step = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
seq = "".join("ACGU"[k] for k in np.random.default_rng(7).integers(0, 4, L))
n = (len(seq) - W) // step + 1
for path in [_lib.LIB_PATH] + sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so"))):
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=path)
    eng.scan(seq, W, step, 0, min(n, 512), 1, _lib.SHUFFLE_DI, 1, raw=True)
    best = {}
    for flags in (0, 1):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter()
            res = eng.scan(seq, W, step, 0, n, 1, _lib.SHUFFLE_DI, 2026, flags, raw=True)
            ts.append(time.perf_counter() - t0)
        best[flags] = min(ts)
        if flags == 0:
            chk = (float(res["ens_dG"].sum()), float(res["ens_div"].sum()), zlib.crc32(res["centroid"].tobytes()),
                   zlib.crc32(res["ens_dG"].tobytes()) ^ zlib.crc32(res["ens_div"].tobytes()))
    print("%-24s W %d step %d n %d: with PF %.1f ms, without %.1f ms -> PF %.1f ms   sums %.6f %.6f crc %08x %08x" % (
        os.path.basename(path), W, step, n, best[0] * 1e3, best[1] * 1e3, (best[0] - best[1]) * 1e3, chk[0], chk[1],
        chk[2], chk[3]), flush=True)
    eng.shutdown()

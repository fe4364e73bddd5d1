"""Where do the waves of the MFE kernel's persistent workgroups land, and does it depend on what ran before?
Runs tools/micro/libplace_probe.so (a grid shaped like the W=120 MFE kernel's) plainly, right after the shuffle kernel,
and after a small MFE launch; per CU: the SIMD of wave 0 of each of its four workgroups."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from collections import Counter
from scanfold_amd import _lib
eng = _lib.get_engine(0)
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libplace_probe.so"))
lib.place_probe.argtypes = [ctypes.c_void_p]
seq = "".join("ACGU"[k] for k in np.random.default_rng(7).integers(0, 4, 30000))
rows = np.frombuffer(b"ACGU", dtype=np.uint8)[np.random.default_rng(1).integers(0, 4, (4096, 120))]
def probe(label):
    out = np.zeros(1024 * 8, dtype=np.uint32)
    rc = lib.place_probe(out.ctypes.data)
    assert rc == 0, rc
    hw, xcc = out[0::2].reshape(1024, 4), out[1::2].reshape(1024, 4) & 0xF
    simd, cu, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 13) & 7
    cuid = (xcc[:, 0].astype(np.int64) << 16) | (se[:, 0] << 8) | cu[:, 0]
    distinct_in_wg = int(sum(len(set(simd[b])) == 4 for b in range(1024)))
    pat = Counter()
    per_cu = {}
    for b in range(1024):
        per_cu.setdefault(int(cuid[b]), []).append(int(simd[b, 0]))
    for v in per_cu.values():
        pat[tuple(sorted(Counter(v).values(), reverse=True))] += 1
    wgs = Counter(len(v) for v in per_cu.values())
    print("%-44s CUs %d, workgroups per CU %s, workgroups with 4 distinct SIMDs %d; wave-0 SIMD multiplicities per CU: %s" % (
        label, len(per_cu), dict(wgs), distinct_in_wg, dict(pat)), flush=True)
probe("probe, first launch"); probe("probe again")
eng.shuffle_windows(seq, 120, 1, 0, 29881, 100, _lib.SHUFFLE_DI, 1); probe("after the full shuffle kernel")
probe("probe again")
eng.shuffle_windows(seq, 120, 1, 0, 29881, 100, _lib.SHUFFLE_DI, 1); eng.mfe_batch(rows[:1024]); probe("after shuffle + 1024-fold MFE launch")
eng.mfe_batch(rows); probe("after a 4096-fold MFE launch")
eng.shuffle_windows(seq, 120, 1, 0, 2000, 100, _lib.SHUFFLE_DI, 1); probe("after a 2000-window shuffle")

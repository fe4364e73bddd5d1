import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib, params
from oracle import oracle
oracle.set_params(params.default_params())
eng = _lib.get_engine(0)
rng = np.random.default_rng(0)
for W in (40, 60, 64, 65, 70, 80, 100, 110, 120, 128, 130, 160, 200, 256):
    n = 512
    arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (n, W))]
    ref = oracle.mfe_batch(arr)
    res = []
    for rep in range(3):
        e = eng.mfe_batch(arr)
        res.append(int((e != ref).sum()))
    bad = np.where(e != ref)[0][:5]
    print(W, res, [(int(e[b]), int(ref[b])) for b in bad], flush=True)

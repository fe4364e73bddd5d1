import sys, os, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scanfold_amd import _lib, params
from oracle import oracle
oracle.set_params(params.default_params())
rng = np.random.default_rng(0)
arrs = {W: np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (256, W))] for W in (40, 60, 120)}
refs = {W: oracle.mfe_batch(a) for W, a in arrs.items()}
for path in [_lib.LIB_PATH] + sorted(glob.glob(os.path.join(ROOT, "tools", "abl_*.so"))):
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=path)
    print(os.path.basename(path), {W: int((eng.mfe_batch(a) != refs[W]).sum()) for W, a in arrs.items()}, flush=True)
    eng.shutdown()

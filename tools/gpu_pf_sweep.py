import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, time
sys.path.insert(0, %r)
import numpy as np
from scanfold_amd import _lib
eng = _lib.get_engine(0)
rng = np.random.default_rng(0)
arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (8192, 120))]
eng.pf_batch(arr[:512])
t0 = time.time(); eng.pf_batch(arr); t1 = time.time()
print("pf 8192 x120: %%.3fs" %% (t1 - t0))
''' % ROOT
for b in (1, 2, 3, 4):
    env = dict(os.environ, SCANFOLD_PF_BLOCKS_PER_CU=str(b))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(b, out.stdout.strip(), out.stderr.strip()[-100:], flush=True)

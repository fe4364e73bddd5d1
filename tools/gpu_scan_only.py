"""One scan of a bench.py workload and nothing else — the process `rocprofv3 --pmc` wraps when bench.py (or a round's
measurement script) wants the counters of the TIMED kernel: the same transcript, window range, shuffles and launch shapes as the
bench step, no warm-up launch (every sf_mfe_fast_kernel dispatch of this process belongs to the step).

    python tools/gpu_scan_only.py [config] [--shuffle di|mono] [--input uniform|viral] [--windows N]

--windows N: only the first N windows (cfg5's step is 33 s; its counters are taken on a slice).  Prints one JSON line:
windows, folds, launches, kernel ms (HIP events on the launch stream), folds/s."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload table and transcript generator only)
from scanfold_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="cfg3", choices=sorted(bench.WORKLOADS))
ap.add_argument("--shuffle", default="")
ap.add_argument("--input", default="uniform")
ap.add_argument("--windows", type=int, default=0)
a = ap.parse_args()
wl = bench.WORKLOADS[a.config]
kind = _lib.SHUFFLE_MONO if (a.shuffle or wl["shuffle"]) == "mono" else _lib.SHUFFLE_DI
seq = bench.synth_transcript(wl["L"], wl["seed"], a.input)
W, step, r = wl["W"], wl["step"], wl["r"]
n_win = (len(seq) - W) // step + 1
if a.windows > 0:
    n_win = min(n_win, a.windows)
path = os.environ.get("SCANFOLD_LIB")
if path:
    _lib._share_hip_runtime_with_torch()
    eng = _lib.Engine(0, lib_path=os.path.join(ROOT, path))
else:
    eng = _lib.Engine(0)
eng.prof_reset()
res = eng.scan(seq, W, step, 0, n_win, r, kind, wl["shuffle_seed"], raw=True)
ms, nl, nf = eng.prof_get()
print(json.dumps({"config": a.config, "windows": n_win, "folds": int(nf), "launches": int(nl), "kernel_ms": ms,
                  "folds_per_s": nf / ms * 1e3 if ms > 0 else None, "energy_checksum": int(res["energies"].sum())}))

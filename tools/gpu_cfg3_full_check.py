"""EVERY window of a bench workload against the CPU engine (oracle/sf_cpu_twin.c, which tests/ pin to the oracle): the GPU scan's
(r+1) energies, structure, centroid and ensemble diversity of all windows — cfg3: 29 881 windows = 3 017 981 energies — compared
with the twin's on the same shuffled rows (the device shuffles, fetched with sf_shuffle_windows).  bench.py checks 64 windows of
every run; this is the full-size check (about 6 minutes of host time on 16 threads).   usage: gpu_cfg3_full_check.py [config] [windows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from scanfold_amd import _lib, params
from oracle import oracle

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
wl = bench.WORKLOADS[cfg]
L, W, step, r = wl["L"], wl["W"], wl["step"], wl["r"]
seq = bench.synth_transcript(L, wl["seed"], "uniform")
n_win = (L - W) // step + 1
if len(sys.argv) > 2:
    n_win = min(n_win, int(sys.argv[2]))
kind = _lib.SHUFFLE_DI if wl["shuffle"] == "di" else _lib.SHUFFLE_MONO
oracle.build(); oracle.set_params(params.default_params())
eng = _lib.Engine(0)
t0 = time.time()
res = eng.scan(seq, W, step, 0, n_win, r, kind, wl["shuffle_seed"])
t_gpu = time.time() - t0
bad = dict(energies=0, structure=0, centroid=0, ens_div=0)
CH = 2048
ascii_of = np.frombuffer(b"NACGU", dtype=np.uint8)
t0 = time.time()
for w0 in range(0, n_win, CH):
    nw = min(CH, n_win - w0)
    rows = ascii_of[eng.shuffle_windows(seq, W, step, w0, nw, r, kind, wl["shuffle_seed"])]
    ref = oracle.twin_scan_windows(rows, nw, r)
    bad["energies"] += int((ref["energies"] != res["energies"][w0:w0 + nw]).sum())
    bad["structure"] += sum(a != b for a, b in zip(ref["structure"], res["structure"][w0:w0 + nw]))
    bad["centroid"] += sum(a != b for a, b in zip(ref["centroid"], res["centroid"][w0:w0 + nw]))
    bad["ens_div"] += int((np.abs(ref["ens_div"] - res["ens_div"][w0:w0 + nw]) > 1e-9).sum())
    print("windows %d..%d checked, mismatches so far %s (%.0f s)" % (w0, w0 + nw - 1, bad, time.time() - t0), flush=True)
print("%s: %d windows, %d energies, W=%d step=%d r=%d %s-shuffles: GPU scan %.2f s, CPU engine %.0f s; mismatches: %s"
      % (cfg, n_win, n_win * (r + 1), W, step, r, wl["shuffle"], t_gpu, time.time() - t0, bad))
sys.exit(1 if any(bad.values()) else 0)

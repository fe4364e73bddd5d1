"""The sharded command line on cfg3 (30 kb, W=120, step=1, r=100): one process against two ranks on the SAME GPU (gloo
gather; the box has one GPU, so the GPU time does not halve — what is measured is the HOST time per rank: z/p-scores, row
formatting and the gather, which the round-2 CLI did for all windows on every rank)."""
import os, subprocess, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
seq = "".join("ACGU"[k] for k in np.random.default_rng(3).integers(0, 4, 30000))
d = tempfile.mkdtemp()
fa = os.path.join(d, "cfg3.fa")
open(fa, "w").write(">cfg3\n" + seq + "\n")
env = dict(os.environ, SCANFOLD_DEVICE="0", SCANFOLD_DIST_BACKEND="gloo")
outs = {}
for gpus in (1, 2, 1, 2):
    out = os.path.join(d, "out%d.tsv" % gpus)
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-m", "scanfold_amd.scan", "-i", fa, "-w", "120", "-s", "1", "-r", "100", "-type", "di",
                        "--seed", "2026", "--timing", "--gpus", str(gpus), "-o", out], cwd=ROOT, env=env, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    print("--gpus %d: rc %d, process wall %.2f s (incl. interpreter + torch start-up)" % (gpus, p.returncode, wall))
    for ln in p.stderr.splitlines():
        if "timing rank" in ln:
            print("   ", ln)
    if p.returncode:
        print(p.stderr[-2000:])
    outs[gpus] = open(out, "rb").read() if os.path.exists(out) else None
print("TSV of --gpus 2 == TSV of --gpus 1:", outs[1] is not None and outs[1] == outs[2], "bytes", outs[1] and len(outs[1]))

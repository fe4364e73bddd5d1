"""What ONE rank of an N-way split does per step, timed on ONE GPU — NOT a scaling measurement (no second GPU, no xGMI, no
RCCL): it answers "does the per-rank step divide by N?" before the driver's multi-GPU run does.

    python tools/gpu_shard_step.py [cfg3|cfg5] [--splits 1,2,4,8] [--reps 5] [--windows N]

For every N in --splits: rank 0's shard of bench.py's workload (dist.shard_range(n_win, 0, N): the first ceil(n_win / N)
windows; every rank's shard has the same size to within one window and a window costs the same wherever it lies) through the
SAME device-resident step as bench.py — eng.scan_dev on the shard, then dist.pack_records on the device (the all-gather itself
is absent: 2.5 MB per rank at cfg3 / 8 ranks, ~20 us of wire time at xGMI rates).  Median of --reps steps after one warm-up,
HIP-event kernel times of the MFE launch, and T(1) / T(N): what an N-GPU run could reach at best if nothing else got in its way.
What does not divide: the persistent grid's tail (1 024 workgroups finish at different times), the partition function's
run-length schedule (a run's first window is a full fold: scanfold_hip.hip, sf_scan's PF runs), launch overheads, pack_records.

--windows N: take only the first N windows of the workload as "the whole" (cfg5's full step is ~29 s).
Prints one JSON line per split and a summary table."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload table and transcript generator only)
import torch  # noqa: E402
from scanfold_amd import _lib, dist as sdist  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("config", nargs="?", default="cfg3", choices=sorted(bench.WORKLOADS))
ap.add_argument("--splits", default="1,2,4,8")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--windows", type=int, default=0)
a = ap.parse_args()
wl = bench.WORKLOADS[a.config]
W, step, r = wl["W"], wl["step"], wl["r"]
kind = _lib.SHUFFLE_DI if wl["shuffle"] == "di" else _lib.SHUFFLE_MONO
seq = bench.synth_transcript(wl["L"], wl["seed"])
n_win = (len(seq) - W) // step + 1
if a.windows > 0:
    n_win = min(n_win, a.windows)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
eng = _lib.Engine(0)
tr = torch.tensor(list(seq.encode()), dtype=torch.uint8, device=dev)
rows = []
for world in [int(x) for x in a.splits.split(",")]:
    lo, hi = sdist.shard_range(n_win, 0, world)
    n_loc, n_pad = hi - lo, sdist.shard_size(n_win, world)
    en = torch.zeros((n_loc, r + 1), dtype=torch.int32, device=dev)
    db = torch.zeros((n_loc, W + 1), dtype=torch.uint8, device=dev)
    cen = torch.zeros((n_loc, W + 1), dtype=torch.uint8, device=dev)
    div = torch.zeros(n_loc, dtype=torch.float64, device=dev)
    dG = torch.zeros(n_loc, dtype=torch.float64, device=dev)

    def one():
        st = torch.cuda.current_stream().cuda_stream
        eng.scan_dev(tr.data_ptr(), len(seq), W, step, lo, n_loc, r, kind, wl["shuffle_seed"], 0, en.data_ptr(),
                     db.data_ptr(), cen.data_ptr(), div.data_ptr(), dG.data_ptr(), st)
        t1 = time.perf_counter()
        rec = sdist.pack_records(torch, W, r, en, db, cen, div, dG, n_pad)
        return rec, t1

    one()
    torch.cuda.synchronize()
    times, mfe_ms, pack_ms = [], [], []
    for _ in range(a.reps):
        eng.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rec, t1 = one()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        times.append(t2 - t0)
        ms, nl, nf = eng.prof_get()
        mfe_ms.append(ms)
        # pack_records alone, timed after the scan has drained
        t3 = time.perf_counter()
        sdist.pack_records(torch, W, r, en, db, cen, div, dG, n_pad)
        torch.cuda.synchronize()
        pack_ms.append((time.perf_counter() - t3) * 1e3)
    med = sorted(times)[len(times) // 2]
    row = {"config": a.config, "ranks": world, "windows_of_rank0": n_loc, "folds_of_rank0": n_loc * (r + 1),
           "step_s_median": med, "step_s_min": min(times), "step_s_max": max(times),
           "mfe_launch_ms_median": sorted(mfe_ms)[len(mfe_ms) // 2], "mfe_launches": int(nl),
           "rest_of_step_ms": med * 1e3 - sorted(mfe_ms)[len(mfe_ms) // 2],
           "pack_records_ms_median": sorted(pack_ms)[len(pack_ms) // 2], "record_bytes": int(rec.numel()),
           "energy_checksum": int(en.sum().item())}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del en, db, cen, div, dG, rec
base = rows[0]
print("\n%s on ONE MI355X (%s): the step of rank 0's shard of an N-way split — not a scaling measurement" % (a.config, eng.device_name()))
print("%5s %9s %11s %10s %10s %10s %12s %14s" % ("ranks", "windows", "step ms", "MFE ms", "rest ms", "pack ms", "T(%d)/T(N)" % base["ranks"],
                                               "of ideal N"))
for row in rows:
    ratio = base["step_s_median"] / row["step_s_median"]
    print("%5d %9d %11.2f %10.2f %10.2f %10.3f %12.2f %13.1f%%" % (
        row["ranks"], row["windows_of_rank0"], row["step_s_median"] * 1e3, row["mfe_launch_ms_median"], row["rest_of_step_ms"],
        row["pack_records_ms_median"], ratio, 100.0 * ratio * base["ranks"] / row["ranks"]))

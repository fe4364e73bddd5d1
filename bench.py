#!/usr/bin/env python3
"""bench.py — windows/sec of the ScanFold-Scan hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path over the workload: for every window of the synthetic transcript,
the native MFE fold + traceback, the partition function (centroid, ensemble diversity), r dinucleotide shuffles
generated on the device and r+1 MFE folds (ScanFold-Scan.py:355-449).  Transcript and all outputs are
resident in HBM when the timed region starts.  With N>1 ranks (one process per GPU, launched by
torch.distributed.run) the windows are split into contiguous ranges and each step ends with ONE RCCL
all-gather of the fixed-size per-window records (scanfold_amd/dist.py); total work is fixed -> "strong".

Workload at N=1: BASELINE.json configs[2] ("30 kb, W=120, step=1, 100 shuffles on 1 MI355X") — the configuration
the metric is quoted on; it fits one GPU.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOAD = dict(name="cfg3: 30 kb synthetic RNA, W=120, step=1, 100 di-shuffles", L=30000, seed=3, W=120, step=1,
                r=100, shuffle="di", shuffle_seed=2026)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def synth_transcript(L, seed):
    return "".join("ACGU"[k] for k in np.random.default_rng(seed).integers(0, 4, L))


def cpu_baseline(eng, seq, W, step, r, kind, seed, budget_s=15.0):
    """The oracle (a CPU port of the ViennaRNA-shaped path; see oracle/sf_oracle.c) on a bounded sample of the
    same workload: per window 1 MFE + traceback, 1 partition function and r+1 MFE folds, one OpenMP thread per
    window on every host core.  Baseline only — never the product."""
    from oracle import oracle
    from scanfold_amd import params
    oracle.build()
    oracle.set_params(params.default_params())
    try:
        cores = len(os.sched_getaffinity(0)) or 1  # the threads this process may actually run on
    except AttributeError:
        cores = os.cpu_count() or 1

    def run(n_win):
        rows = np.frombuffer(b"NACGU", dtype=np.uint8)[eng.shuffle_windows(seq, W, step, 0, n_win, r, kind, seed)]
        t0 = time.perf_counter()
        oracle.scan_windows(rows, n_win, r, nthreads=cores)
        return time.perf_counter() - t0
    n0 = max(cores, 8)
    t0 = run(n0)
    n = int(max(n0, min(20000, n0 * budget_s / max(t0, 1e-6))))
    n = (n // cores) * cores or n0
    t = run(n)
    return dict(value=n / t, unit="windows/s", cores=cores, kind="port",
                sample="first %d windows of the workload, one OpenMP thread per window on %d threads (each window: "
                       "1 MFE + traceback, 1 partition function, %d MFE folds), %.1f s wall; oracle/sf_oracle.c, "
                       "not ViennaRNA (absent)" % (n, cores, r + 1, t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from scanfold_amd import _lib, dist as sdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SCANFOLD_BENCH_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): take the RCCL path with a single
    # rank, to check process-group set-up, record packing and the all-gather on a one-GPU box
    use_dist = world > 1 or os.environ.get("SCANFOLD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=dev)
    eng = _lib.Engine(device=local_rank)

    wl = WORKLOAD
    W, step, r = wl["W"], wl["step"], wl["r"]
    kind = _lib.SHUFFLE_DI if wl["shuffle"] == "di" else _lib.SHUFFLE_MONO
    seq = synth_transcript(wl["L"], wl["seed"])
    n_win = (len(seq) - W) // step + 1
    lo, hi = sdist.shard_range(n_win, rank, world)
    n_loc = hi - lo
    n_pad = sdist.shard_size(n_win, world)

    tr = torch.tensor(list(seq.encode()), dtype=torch.uint8, device=dev)
    en = torch.zeros((max(n_loc, 1), r + 1), dtype=torch.int32, device=dev)
    db = torch.zeros((max(n_loc, 1), W + 1), dtype=torch.uint8, device=dev)
    cen = torch.zeros((max(n_loc, 1), W + 1), dtype=torch.uint8, device=dev)
    div = torch.zeros(max(n_loc, 1), dtype=torch.float64, device=dev)
    dG = torch.zeros(max(n_loc, 1), dtype=torch.float64, device=dev)

    def step_fn():
        st = torch.cuda.current_stream().cuda_stream
        eng.scan_dev(tr.data_ptr(), len(seq), W, step, lo, n_loc, r, kind, wl["shuffle_seed"], 0, en.data_ptr(),
                     db.data_ptr(), cen.data_ptr(), div.data_ptr(), dG.data_ptr(), st)
        if use_dist:
            rec = sdist.pack_records(torch, W, r, en[:n_loc], db[:n_loc], cen[:n_loc], div[:n_loc], dG[:n_loc], n_pad)
            return sdist.gather_records(rec, world)
        return None

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_fn()
    sync()
    eng.prof_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn()
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms, launches, folds = eng.prof_get()

    if rank == 0:
        value = n_win * args.steps / elapsed
        bytes_per_fold = W + 4  # W one-byte nucleotides in, one int32 energy out
        avg_ms = kern_ms / max(launches, 1)
        folds_per_launch = folds / max(launches, 1)
        achieved = folds_per_launch * bytes_per_fold / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "windows/sec (W=120, step=1, 100 shuffles)",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32 (int16 LDS storage) energies; f64 partition function",
            "data": "synthetic",
            "config": {"workload": wl["name"], "L": wl["L"], "W": W, "step": step, "shuffles": r,
                       "shuffle_type": wl["shuffle"], "windows": n_win, "mfe_folds_per_step": n_win * (r + 1),
                       "parallelism": "windows sharded over %d rank(s), one all-gather per step" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "sf_mfe_fast_kernel", "avg_launch_ms": avg_ms, "launches": launches,
                         "folds_per_launch": folds_per_launch, "algorithmic_bytes_per_fold": bytes_per_fold,
                         "note": "integer min-plus DP on LDS-resident tables: LDS/VALU-bound by construction "
                                 "(SURVEY.md F9); HBM fraction reported as the north star asks",
                         "mfe_kernel_share_of_step": (kern_ms * 1e-3) / elapsed},
            "device": eng.device_name(),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eng, seq, W, step, r, kind, wl["shuffle_seed"])
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

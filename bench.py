#!/usr/bin/env python3
"""bench.py — windows/sec of the ScanFold-Scan hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path over the workload: for every window of the synthetic transcript,
the native MFE fold + traceback, the partition function (centroid, ensemble diversity), r dinucleotide shuffles
generated on the device and r+1 MFE folds (ScanFold-Scan.py:355-449).  Transcript and all outputs are
resident in HBM when the timed region starts.  With N>1 ranks (one process per GPU) the windows are split into
contiguous ranges and each step ends with ONE RCCL all-gather of the fixed-size per-window records
(scanfold_amd/dist.py); total work is fixed -> "strong".

`python bench.py --gpus N` works both ways: under `python -m torch.distributed.run --nproc-per-node N` (RANK /
WORLD_SIZE in the environment) it is one rank; started plainly with N > 1 it starts that launcher as a CHILD
process (never exec), relays the child's JSON line and exits with its code.

Workload at N=1: BASELINE.json configs[2] ("30 kb, W=120, step=1, 100 shuffles on 1 MI355X") — the configuration
the metric is quoted on; it fits one GPU.

After the timed region rank 0 checks 64 windows of the last step against the oracle ("verified_windows"), and at
N=1 adds an "e2e" block (the CLI's whole path: H2D, kernels, D2H, z/p-scores, TSV rows) and the CPU baseline.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[0] / configs[1]: the small parity shapes — here so that the bench's own path (device-resident
    # scan, record packing, the RCCL gather, the self-check) can run in seconds inside the GPU test suite
    "cfg1": dict(name="cfg1: 1 kb synthetic RNA, W=120, step=40, 10 di-shuffles", L=1000, seed=1, W=120, step=40, r=10,
                 shuffle="di", shuffle_seed=2026, metric="windows/sec (W=120, step=40, 10 shuffles)", verify=23,
                 counters="profiles/r05/mfe_counters.json"),
    "cfg2": dict(name="cfg2: 10 kb synthetic RNA, W=120, step=10, 30 di-shuffles", L=10000, seed=2, W=120, step=10, r=30,
                 shuffle="di", shuffle_seed=2026, metric="windows/sec (W=120, step=10, 30 shuffles)", verify=64,
                 counters="profiles/r05/mfe_counters.json"),
    # BASELINE.json configs[2] — the configuration the metric is quoted on; the default
    "cfg3": dict(name="cfg3: 30 kb synthetic RNA, W=120, step=1, 100 di-shuffles", L=30000, seed=3, W=120, step=1, r=100,
                 shuffle="di", shuffle_seed=2026, metric="windows/sec (W=120, step=1, 100 shuffles)", verify=64,
                 counters="profiles/r05/mfe_counters.json"),
    # BASELINE.json configs[4] on the GPUs given (`--config cfg5`; one step is ~36 s on one MI355X)
    "cfg5": dict(name="cfg5: 30 kb synthetic RNA, W=200, step=1, 1000 di-shuffles + partition function", L=30000, seed=3,
                 W=200, step=1, r=1000, shuffle="di", shuffle_seed=2026,
                 metric="windows/sec (W=200, step=1, 1000 shuffles, partition function)", verify=32, verify_engine="twin",
                 counters="profiles/r05/cfg5_mfe_counters.json"),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
EXIT_NEED_GPUS = 3


def synth_transcript(L, seed, kind="uniform"):
    """uniform: i.i.d. over ACGU (SURVEY.md 8d's generator).  viral: 8d's secondary input — 60 % A+U background with
    an 80-nt hairpin (38-bp GC-rich stem, GAAA loop) planted every 5 kb, so that some native windows are much more
    stable than their shuffles (z-scores far from 0) — the workload shape a real scan has."""
    import numpy as np
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return "".join("ACGU"[k] for k in rng.integers(0, 4, L))
    if kind != "viral":
        raise ValueError("input kind must be 'uniform' or 'viral'")
    bg = ["ACGU"[k] for k in rng.choice(4, L, p=[0.3, 0.2, 0.2, 0.3])]
    comp = str.maketrans("ACGU", "UGCA")
    for pos in range(1000, L - 100, 5000):
        stem = "".join("ACGU"[k] for k in rng.choice(4, 38, p=[0.15, 0.35, 0.35, 0.15]))
        bg[pos:pos + 80] = list(stem + "GAAA" + stem[::-1].translate(comp))
    return "".join(bg)


def relaunch_under_torchrun(args):
    """--gpus N without a launcher around us: be the launcher's parent.  Nothing in this process has touched the
    GPU (torch is not even imported yet)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    cmd += ["--config", args.config, "--shuffle", args.shuffle or "", "--input", args.input]
    if args.no_live_counters:
        cmd.append("--no-live-counters")
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif proc.returncode == 0:
        print("bench.py: the %d-rank run produced no result line" % args.gpus, file=sys.stderr)
        return 1
    return proc.returncode


def host_cores():
    """(threads this process may run on, physical cores among them, sockets) — physical = distinct (package, core)."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cpus = list(range(os.cpu_count() or 1))
    phys, socks = set(), set()
    for c in cpus:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            pkg = open(base + "physical_package_id").read().strip()
            core = open(base + "core_id").read().strip()
            phys.add((pkg, core))
            socks.add(pkg)
        except OSError:
            phys.add(("?", str(c)))
    return len(cpus), len(phys), max(len(socks), 1)


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: the GPU
    boxes show 256 hardware threads but cap the container at 16 CPUs — more threads than that only take turns."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(round(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, int(round(q / p)))
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(seq, W, step, r, kind, seed, budget_s=12.0):
    """The CPU engine (oracle/sf_cpu_twin.c when built, else the checker oracle/sf_oracle.c) on a bounded sample of
    the same workload: per window 1 MFE + traceback, 1 partition function and r+1 MFE folds, one OpenMP thread per
    window, threads = physical cores (capped by the container's CPU quota).  TWO builds of the same sources are timed on
    the same sample: the portable one (`-O3`, what travels with the repository) and one made NOW, on this host, the way a
    user would build for it (`-O3 -march=native`, oracle.build_native); `value` is the FASTER of the two.
    Baseline only — never the product, never ViennaRNA (absent)."""
    import numpy as np
    from oracle import oracle
    from scanfold_amd import params
    oracle.build()
    pset = params.default_params()
    oracle.set_params(pset)
    threads, phys, socks = host_cores()
    quota = cpu_quota()
    cores = phys if quota is None else min(phys, quota)
    engine = "oracle/sf_oracle.c (the parity checker: O(n^4) outside pass, allocations per fold)"
    twin = hasattr(oracle, "twin_available") and oracle.twin_available()
    if twin:
        engine = ("oracle/sf_cpu_twin.c (per-thread workspaces, pair-type matrix, dense interior loops with pre-added "
                  "mismatch view, vectorisable multiloop split, O(n^3) outside pass)")
    native, native_err = None, None
    if twin:
        try:
            native = oracle.lib_native()
            oracle.set_params(pset, L=native)
        except Exception as e:  # no compiler on the box, or the build failed: the portable build stands alone
            native, native_err = None, repr(e)

    def run(n_win, L=None):
        rows = np.frombuffer(b"NACGU", dtype=np.uint8)[oracle.shuffle_windows(seq, W, step, 0, n_win, r, kind, seed)]
        t0 = time.perf_counter()
        if twin:
            res = oracle.twin_scan_windows(rows, n_win, r, nthreads=cores, L=L)
        else:
            res = oracle.scan_windows(rows, n_win, r, nthreads=cores)
        return time.perf_counter() - t0, res
    n0 = max(cores, 8)
    t0, _ = run(n0)
    n = int(max(n0, min(20000, n0 * budget_s / max(t0, 1e-6))))
    n = (n // cores) * cores or n0
    t, res = run(n)
    builds = [{"flags": oracle.BASE_FLAGS, "windows_per_s": n / t, "per_core_windows_per_s": n / t / cores, "wall_s": t,
               "built": "in the build container, travels with the tree (oracle/Makefile CFLAGS)"}]
    if native is not None:
        tn, resn = run(n, L=native)
        same = bool((res["energies"] == resn["energies"]).all() and res["structure"] == resn["structure"]
                    and res["centroid"] == resn["centroid"])
        builds.append({"flags": oracle.NATIVE_FLAGS, "windows_per_s": n / tn, "per_core_windows_per_s": n / tn / cores,
                       "wall_s": tn, "built": "on this host just now (oracle.build_native; oracle/Makefile NATIVE_FLAGS)",
                       "cpu": oracle._cpu_stamp().split(" | ")[0], "results_equal_the_portable_build": same,
                       "speedup_over_portable": t / tn})
    best = max(builds, key=lambda b: b["windows_per_s"])
    try:
        overhead = reference_python_overhead(seq, W, r, windows=4)
    except Exception as e:  # never let the side measurement break the bench line
        overhead = {"error": repr(e)}
    out = dict(value=best["windows_per_s"], unit="windows/s", cores=cores, kind="port", flags=best["flags"],
               per_core=best["per_core_windows_per_s"], builds=builds, reference_python_overhead=overhead,
               host={"hardware_threads": threads, "physical_cores": phys, "sockets": socks, "cgroup_cpu_quota": quota},
               sample="first %d windows of the workload, one OpenMP thread per window on %d threads (%s; the host has "
                      "%d physical cores in %d socket(s), %d hardware threads); each window: 1 MFE + traceback, 1 "
                      "partition function, %d MFE folds; %.1f s wall per build (%d build(s) timed on the same sample, `value` = the "
                      "faster: %s); engine: %s — not ViennaRNA (absent)"
                      % (n, cores, "the container's CPU quota" if quota is not None and quota < phys else
                         "one per physical core", phys, socks, threads, r + 1, best["wall_s"], len(builds), best["flags"], engine))
    if native_err:
        out["native_build_error"] = native_err
    return out


def _noop(x):
    return x


def live_counters_per_fold(config, shuffle, input_kind, groups, windows=0, timeout_s=240):
    """Hardware counters of the dominant kernel per fold, measured NOW on the workload that was just timed: one
    `rocprofv3 --pmc <group>` pass per group (separate runs, as MI355X_MICROARCH.md prescribes) of tools/gpu_scan_only.py —
    ONE scan of the same transcript with the same shuffles, no warm-up launch — as CHILD processes (this process keeps the
    GPU; it is idle meanwhile).  Every sf_mfe_fast_kernel dispatch of the child belongs to the step, so the sums are the
    step's.  -> ({counter: value per fold, "_folds": n, "_launches": n}, None) or (None, why not)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tool = os.path.join(ROOT, "tools", "gpu_scan_only.py")
    targs = [config, "--input", input_kind] + (["--shuffle", shuffle] if shuffle else []) + (["--windows", str(windows)] if windows else [])
    vals, folds, launches = {}, None, None
    tmp = tempfile.mkdtemp(prefix="sf_pmc_", dir="/tmp")
    try:
        for k, group in enumerate(groups):
            out = os.path.join(tmp, "g%d" % k)
            try:
                p = subprocess.run([exe, "--pmc"] + group.split() + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
                                    sys.executable, tool] + targs, cwd="/tmp",
                                   env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   text=True, timeout=timeout_s)
            except Exception as e:
                return None, "rocprofv3 --pmc %s: %r" % (group, e)
            for ln in p.stdout.splitlines():
                if ln.startswith("{") and '"folds"' in ln:
                    j = json.loads(ln)
                    folds, launches = j["folds"], j["launches"]
            got, disp = {}, set()
            for f in glob.glob(os.path.join(out, "*", "*counter_collection.csv")):
                for row in csv.DictReader(open(f)):
                    if "sf_mfe_fast_kernel" in row["Kernel_Name"]:
                        got[row["Counter_Name"]] = got.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                        disp.add(row.get("Dispatch_Id"))
            missing = [c for c in group.split() if got.get(c, 0.0) <= 0.0]
            if missing or not folds:
                return None, "rocprofv3 --pmc %s gave no rows for %s (rc %d: %s)" % (group, missing, p.returncode, p.stderr[-200:])
            if launches and len(disp) != launches:
                return None, "rocprofv3 --pmc %s: %d dispatches of the kernel for %d launches" % (group, len(disp), launches)
            vals.update(got)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {k: v / folds for k, v in vals.items()}
    out["_folds"], out["_launches"] = folds, launches
    return out, None


def calibrated_unit_fractions(sec, W, folds_per_launch, launch_ms, eng=None, n_cu=256):
    """How busy the vector ALUs and the LDS were during the timed launch: instructions per fold (SQ counters) x the folds of
    a launch x the MEASURED cost of an instruction of this kernel's mix at its occupancy (profiles/r05/mfe_issue_rates.json:
    tools/micro/issue_rates.hip, valu_classes.hip; vector mix from the ISA of the hot blocks) / the launch's duration by HIP
    events.  (Rounds 1-3 assumed 4 cycles per vector instruction; the round-3 review held 2 against it.  Measured: the
    packed int16 / min / bit-select instructions are half-rate, 1.8-1.95 ns per wave64 instruction and SIMD, 32-bit adds
    and moves full-rate, 1.05 ns; this kernel's mix averages 1.72 ns.)"""
    path = os.path.join(ROOT, "profiles", "r05", "mfe_issue_rates.json")
    try:
        rates = json.load(open(path))
        name = eng.device_name() if eng is not None else ""
        if "CUs" in name:
            n_cu = int(name.split(",")[-1].split()[0])
    except Exception as e:
        return {"calibration": "unavailable: %r" % e}
    if launch_ms <= 0:
        return {"calibration": "no launch time"}
    launch_ns = launch_ms * 1e6
    v_ns = rates["valu_ns_per_inst_per_simd"].get(str(W), rates["valu_ns_per_inst_per_simd"]["default"])
    valu = sec["valu_insts_per_fold"] * folds_per_launch / (4.0 * n_cu) * v_ns / launch_ns
    salu = sec["salu_insts_per_fold"] * folds_per_launch / (4.0 * n_cu) * rates["salu_ns_per_inst_per_simd"] / launch_ns
    lds = sec["lds_idx_active_per_fold"] * folds_per_launch / n_cu * rates["lds_ns_per_idx_active_unit_per_cu"] / launch_ns
    return {"valu_issue_frac": valu, "lds_frac": lds, "salu_issue_frac": salu,
            "binding_unit": "valu" if valu >= lds else "lds",
            "measured_peaks": {"valu_ns_per_wave_inst_per_simd": v_ns,
                               "valu_full_rate_ns": rates["valu_full_rate_ns"], "valu_half_rate_ns": rates["valu_half_rate_ns"],
                               "salu_ns_per_wave_inst_per_simd": rates["salu_ns_per_inst_per_simd"],
                               "lds_ns_per_idx_active_unit_per_cu": rates["lds_ns_per_idx_active_unit_per_cu"],
                               "lds_ns_per_inst_per_cu": rates["lds_ns_per_inst_per_cu"], "n_cu": n_cu},
            "definition": "frac = instructions (or LDS index-active units) of one launch per SIMD (per CU) x measured ns per "
                          "instruction at four waves per SIMD / launch duration; 1.0 = that unit issues back to back",
            "calibration": "profiles/r05/mfe_issue_rates.json"}


def reference_python_overhead(seq, W, r, windows=6):
    """What the reference adds per window on top of its ViennaRNA calls (ScanFold-Scan.py:73-77,256,269-274): r pure
    Python dinucleotide shuffles in the parent (scanfold_amd.functions.dinuclShuffle is the reference's function draw
    for draw, tests/test_golden_host.py) and one ProcessPoolExecutor(12) created, fed r+1 items and torn down.
    Seconds per window, measured on a few windows; lets a reference-shaped figure be reconstructed from a fold rate."""
    import random
    from concurrent.futures import ProcessPoolExecutor
    from scanfold_amd import functions as sff
    random.seed(1)
    t0 = time.perf_counter()
    for w in range(windows):
        frag = seq[w * 50:w * 50 + W]
        for _ in range(r):
            sff.dinuclShuffle(frag)
    t_shuffle = (time.perf_counter() - t0) / windows
    t0 = time.perf_counter()
    for w in range(windows):
        with ProcessPoolExecutor(12) as ex:
            list(ex.map(_noop, range(r + 1)))
    t_pool = (time.perf_counter() - t0) / windows
    return {"python_dinucl_shuffles_s_per_window": t_shuffle, "process_pool_12_s_per_window": t_pool,
            "windows_per_s_if_folds_were_free": 1.0 / (t_shuffle + t_pool)}


def verify_indices(n_loc, n_check):
    """The windows verify_sample checks: n_check of n_loc, evenly spread, first and last included."""
    import numpy as np
    n_check = min(n_check, n_loc)
    if n_check <= 0:
        return np.zeros(0, dtype=np.int64)
    return np.unique(np.linspace(0, n_loc - 1, n_check).astype(np.int64))


def verify_sample(seq, W, step, r, kind, seed, lo, n_loc, en, db, cen, div, n_check=64, paramset=None, engine="checker"):
    """Compare n_check windows of the LAST timed step (device tensors of rank 0's shard) with the oracle: every one
    of the r+1 energies on the oracle's own shuffles, structure, centroid, ensemble diversity.  Returns
    (windows checked, mismatching windows).  engine: "checker" = oracle/sf_oracle.c (O(n^4) outside pass: seconds per 200-mer
    window with 1001 folds), "twin" = oracle/sf_cpu_twin.c, the fast CPU engine that tests/test_oracle.py holds equal to the
    checker — what cfg5's 32 windows x 1001 folds of 200-mers are checked with."""
    import numpy as np
    from oracle import oracle
    from scanfold_amd import params
    oracle.build()
    oracle.set_params(paramset if paramset is not None else params.default_params())
    idx = verify_indices(n_loc, n_check)
    if len(idx) == 0:
        return 0, 0
    e_dev = en[idx].cpu().numpy()
    db_dev = db[idx].cpu().numpy()
    cen_dev = cen[idx].cpu().numpy()
    div_dev = div[idx].cpu().numpy()
    rows = np.concatenate([oracle.shuffle_windows(seq, W, step, lo + int(w), 1, r, kind, seed) for w in idx])
    ascii_rows = np.frombuffer(b"NACGU", dtype=np.uint8)[rows]
    if engine == "twin" and oracle.twin_available():
        ref = oracle.twin_scan_windows(ascii_rows, len(idx), r)
    else:
        ref = oracle.scan_windows(ascii_rows, len(idx), r)
    bad = 0
    for k in range(len(idx)):
        ok = (ref["energies"][k] == e_dev[k]).all() and ref["structure"][k] == bytes(db_dev[k, :W]).decode() \
            and ref["centroid"][k] == bytes(cen_dev[k, :W]).decode() and abs(ref["ens_div"][k] - div_dev[k]) < 1e-8
        bad += 0 if ok else 1
    return len(idx), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="take roofline.traffic from the committed profiles/ file instead of two live rocprofv3 --pmc child "
                         "runs (use when bench.py itself runs under rocprofv3)")
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="cfg3",
                    help="cfg3 (default, the metric's own configuration), cfg5 (W=200, r=1000, partition function), or the "
                         "small parity shapes cfg1 / cfg2")
    ap.add_argument("--shuffle", default="", help="di (the north star's; default) or mono (SURVEY.md 8d's secondary run)")
    ap.add_argument("--input", choices=("uniform", "viral"), default="uniform",
                    help="uniform i.i.d. ACGU (default) or SURVEY.md 8d's viral-like input (60 %% AU + planted hairpins)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 2 if args.config == "cfg3" else 1
    if args.warmup is None:
        args.warmup = 1 if args.config == "cfg3" else 0
    if args.gpus > 1 and "RANK" not in os.environ:
        return relaunch_under_torchrun(args)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        return 2
    n_dev = torch.cuda.device_count()  # does not initialise the GPU
    # SCANFOLD_DIST_BACKEND=gloo + SCANFOLD_DEVICE=k: several ranks on ONE GPU (RCCL refuses that), the collective staged
    # through the host — how the -m gpu suite runs this file with world > 1 on a one-GPU box (tests/test_gpu_paths.py)
    backend = os.environ.get("SCANFOLD_DIST_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        print("bench.py: SCANFOLD_DIST_BACKEND must be nccl or gloo", file=sys.stderr)
        return 2
    dev_index = local_rank
    if backend == "gloo" and os.environ.get("SCANFOLD_DEVICE", "") != "":
        dev_index = int(os.environ["SCANFOLD_DEVICE"])
    elif n_dev < world:
        print("bench.py rank %d: --gpus %d needs %d GPUs, %d GPU(s) visible" % (rank, args.gpus, world, n_dev),
              file=sys.stderr)
        return EXIT_NEED_GPUS
    if dev_index >= n_dev:
        print("bench.py rank %d: device %d of %d GPU(s) visible" % (rank, dev_index, n_dev), file=sys.stderr)
        return EXIT_NEED_GPUS
    from scanfold_amd import _lib, dist as sdist
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # SCANFOLD_BENCH_FORCE_DIST=1 (under torch.distributed.run --nproc-per-node 1): take the RCCL path with a single
    # rank — process group, record packing and a one-rank all_gather_into_tensor — on a one-GPU box
    force_dist = os.environ.get("SCANFOLD_BENCH_FORCE_DIST") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    coll_dev = dev if backend == "nccl" else torch.device("cpu")  # where the small collectives' tensors live
    eng = _lib.Engine(device=dev_index)

    wl = dict(WORKLOADS[args.config])
    if args.shuffle:
        if args.shuffle not in ("di", "mono"):
            print("bench.py: --shuffle must be di or mono", file=sys.stderr)
            return 2
        wl["name"] = wl["name"].replace("%s-shuffles" % wl["shuffle"], "%s-shuffles" % args.shuffle)
        wl["shuffle"] = args.shuffle
    if args.input != "uniform":
        wl["name"] = wl["name"].replace("synthetic RNA", "synthetic %s-like RNA" % args.input)
    W, step, r = wl["W"], wl["step"], wl["r"]
    kind = _lib.SHUFFLE_DI if wl["shuffle"] == "di" else _lib.SHUFFLE_MONO
    seq = synth_transcript(wl["L"], wl["seed"], args.input)
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
            return sdist.gather_records(rec, world, force=force_dist)
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
    gathered = None
    for _ in range(args.steps):
        gathered = step_fn()
    sync()
    elapsed = time.perf_counter() - t0
    rank_elapsed = [elapsed]
    if use_dist:
        mine_t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        every = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(every, mine_t)
        rank_elapsed = [float(x) for x in every.cpu()]
        t = mine_t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms, launches, folds = eng.prof_get()
    dev_status = eng.last_status() if hasattr(eng, "last_status") else 0

    rc = 0
    if rank == 0:
        value = n_win * args.steps / elapsed
        bytes_per_fold = W + 4  # W one-byte nucleotides in, one int32 energy out
        avg_ms = kern_ms / max(launches, 1)
        folds_per_launch = folds / max(launches, 1)
        achieved = folds_per_launch * bytes_per_fold / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, secondary, counters_src = None, None, None
        for cand in (wl["counters"],):
            path = os.path.join(ROOT, cand)
            if os.path.exists(path):
                try:
                    j = json.load(open(path))
                    traffic = j.get("hbm_bytes_per_launch")
                    if traffic is None and j.get("hbm_bytes_per_fold") is not None:
                        traffic = j["hbm_bytes_per_fold"] * folds_per_launch
                    secondary = j.get("secondary")
                    counters_src = cand
                    break
                except Exception:
                    pass
        if secondary and "valu_issue_frac" not in secondary and "lds_idx_active_per_fold" in secondary:
            secondary = dict(secondary)
            secondary.update(calibrated_unit_fractions(secondary, W, folds_per_launch, avg_ms, eng))
        traffic_src = counters_src and (counters_src + " (rocprofv3 --pmc passes of an earlier run, committed)")
        secondary_src = counters_src and (counters_src + " (SQ counter passes of an earlier run, committed)")
        if world == 1 and not args.no_live_counters:
            # cfg5's step is half a minute: its counters are taken on the first 2 000 windows of the same scan
            live, why = live_counters_per_fold(args.config, args.shuffle, args.input,
                                               ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE",
                                                "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"],
                                               windows=(2000 if W > 128 else 0))
            if live is not None:
                how = ("live: rocprofv3 --pmc child runs of tools/gpu_scan_only.py %s (ONE scan of the timed workload, no warm-up: "
                       "%d sf_mfe_fast_kernel launch(es), %d folds) after the timed region, one counter group per run"
                       % (args.config, live["_launches"], live["_folds"]))
                # gfx950 reports half of a read: bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB; L2 <-> fabric, Infinity-Cache hits included
                traffic = (2.0 * live["FETCH_SIZE"] + live["WRITE_SIZE"]) * 1024.0 * folds_per_launch
                traffic_src = how + "; (2 x FETCH_SIZE + WRITE_SIZE) KB per fold x the folds of one launch"
                wc = live["SQ_WAVE_CYCLES"]
                secondary = {"lanes_active_of_64": live["SQ_THREAD_CYCLES_VALU"] / max(live["SQ_ACTIVE_INST_VALU"], 1.0),
                             "waves_parked": live["SQ_WAIT_ANY"] / wc,
                             "valu_insts_per_fold": live["SQ_INSTS_VALU"], "lds_insts_per_fold": live["SQ_INSTS_LDS"],
                             "salu_insts_per_fold": live["SQ_INSTS_SALU"], "lds_idx_active_per_fold": live["SQ_LDS_IDX_ACTIVE"]}
                secondary_src = how
                secondary.update(calibrated_unit_fractions(secondary, W, folds_per_launch, avg_ms, eng))
            else:
                traffic_src = (traffic_src or "none") + "; live measurement failed: " + why
        gather_check = None
        if use_dist and gathered is not None:
            # what the RCCL all-gather delivered, un-padded and put back into window order, against (i) this rank's own
            # device tensors for its shard and (ii) the oracle on windows spread over EVERY rank's shard (below)
            merged = sdist.merge_shards(gathered, n_win, world, W, r)
            mine = slice(lo, hi)
            same = bool((merged["energies"][mine] == en[:n_loc].cpu().numpy()).all()
                        and (merged["structure"][mine] == db[:n_loc].cpu().numpy()).all()
                        and (merged["centroid"][mine] == cen[:n_loc].cpu().numpy()).all()
                        and (merged["ens_div"][mine] == div[:n_loc].cpu().numpy()).all()
                        and (merged["ens_dG"][mine] == dG[:n_loc].cpu().numpy()).all())
            gather_check = {"backend": dist.get_backend(), "ranks": world, "windows": n_win,
                            "bytes_per_rank": int(gathered.shape[0] // world * gathered.shape[1]),
                            "rank0_shard_equals_its_device_tensors": same,
                            "shard_ranges": [list(sdist.shard_range(n_win, k, world)) for k in range(world)],
                            "rank_elapsed_s": rank_elapsed, "elapsed_is_max_over_ranks": elapsed == max(rank_elapsed)}
            # verify on the gathered records of all shards: torch tensors on the host in the layout verify_sample reads
            v_en, v_db = torch.from_numpy(merged["energies"]), torch.from_numpy(merged["structure"])
            v_cen, v_div = torch.from_numpy(merged["centroid"]), torch.from_numpy(merged["ens_div"])
            checked, bad = verify_sample(seq, W, step, r, kind, wl["shuffle_seed"], 0, n_win, v_en, v_db, v_cen, v_div,
                                         wl["verify"], engine=wl.get("verify_engine", "checker"))
            picked = verify_indices(n_win, wl["verify"])
            gather_check["verified_windows_per_shard"] = [int(((picked >= a) & (picked < b)).sum())
                                                          for a, b in gather_check["shard_ranges"]]
            if not same:
                bad += 1
        else:
            checked, bad = verify_sample(seq, W, step, r, kind, wl["shuffle_seed"], lo, n_loc, en, db, cen, div, wl["verify"],
                                         engine=wl.get("verify_engine", "checker"))
        algorithmic_bytes = folds_per_launch * bytes_per_fold
        out = {
            "metric": wl["metric"],
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32 (int16 LDS storage) energies; f64 partition function",
            "data": "synthetic",
            "config": {"workload": wl["name"], "L": wl["L"], "W": W, "step": step, "shuffles": r,
                       "shuffle_type": wl["shuffle"], "windows": n_win, "mfe_folds_per_step": n_win * (r + 1),
                       "parallelism": "windows sharded over %d rank(s), one all-gather per step" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algorithmic_bytes,
                         "traffic_over_algorithmic": (traffic / algorithmic_bytes) if traffic and algorithmic_bytes else None,
                         "traffic_source": traffic_src,
                         "secondary_source": secondary_src,
                         "kernel": "sf_mfe_fast_kernel", "avg_launch_ms": avg_ms, "launches": launches,
                         "folds_per_launch": folds_per_launch, "algorithmic_bytes_per_fold": bytes_per_fold,
                         "note": "integer min-plus DP on LDS-resident tables: LDS/VALU-bound by construction "
                                 "(SURVEY.md F9); HBM fraction reported as the north star asks; the unit that bounds "
                                 "the kernel is in `secondary`",
                         "secondary": secondary,
                         "mfe_kernel_share_of_step": (kern_ms * 1e-3) / elapsed},
            "verified_windows": checked, "verified_mismatches": bad, "device_status": dev_status,
            "gather_check": gather_check,
            "verified_against": "oracle (%s + sf_shuffle_oracle.c): all %d energies, structure, centroid, "
                                "ensemble diversity of %d windows of the last timed step%s"
                                % ("sf_cpu_twin.c, the CPU engine tests/test_oracle.py holds equal to the checker sf_oracle.c,"
                                   if wl.get("verify_engine") == "twin" else "sf_oracle.c",
                                   r + 1, checked, " (taken from the gathered records of all ranks)" if gather_check else ""),
            "params": eng.params.source and os.path.basename(eng.params.source),
            "device": eng.device_name(),
        }
        if bad or dev_status:
            rc = 4
        if world == 1:
            from scanfold_amd import scan as scanmod
            times, nbytes = [], 0
            n_e2e = 5 if args.config == "cfg3" else 1
            if n_e2e > 1:
                scanmod.scan_record(seq, W, step, r, wl["shuffle"], 37, eng, seed=wl["shuffle_seed"])  # warm-up
            for _ in range(n_e2e):
                t0 = time.perf_counter()
                rows = scanmod.scan_record(seq, W, step, r, wl["shuffle"], 37, eng, seed=wl["shuffle_seed"])
                times.append(time.perf_counter() - t0)
                nbytes = sum(len(x) for x in rows)
            med = sorted(times)[len(times) // 2]
            out["e2e"] = {"windows_per_s": n_win / med, "median_s": med, "runs": len(times), "min_s": min(times),
                          "max_s": max(times), "tsv_bytes": nbytes, "ratio_to_device_resident": (n_win / med) / value,
                          "what": "scan.scan_record(): transcript H2D, all kernels, D2H, z/p-scores, TSV rows; host "
                                  "formatting of chunk k overlaps the GPU on chunk k+1 (engine calls of %s windows)"
                                  % " + ".join(str(nw) for _, nw in scanmod._chunk_bounds(0, n_win))}
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(seq, W, step, r, kind, wl["shuffle_seed"])
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())

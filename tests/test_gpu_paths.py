"""Paths of the product that had only run in builder-side scripts, now under the driver's `-m gpu` run:

* the RCCL exchange — `bench.py` as ONE rank under torch.distributed.run with SCANFOLD_BENCH_FORCE_DIST=1: process group on
  the "nccl" backend, record packing on the device, all_gather_into_tensor, un-padding, and the oracle check on what the
  gather delivered (north_star: "a single RCCL gather over xGMI at the end");
* a stretch of 120 x N inside a transcript through scan_record on the HIP engine (ScanFold-Scan.py:374-380: the literal
  row; the device dinucleotide shuffle of a one-symbol window; the partition function of a window that cannot pair);
* `-type mono` through the command line against the shuffle oracle on BASELINE configs 1 and 2 (ScanFold-Scan.py:273-274);
* the slack the short-diagonal cell code over-reads, poisoned (SCANFOLD_MFE_POISON): energies must not move;
* the -t path against the independent Python model of tests/py_model.py;
* the per-XCD placement of the MFE scratch tables and of the parked partition-function state with grids that are not multiples of eight.
Nothing here reads /root/reference."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from scanfold_amd import _lib
from scanfold_amd import scan as scanmod
from conftest import random_seqs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ascii_rows(codes):
    return np.frombuffer(b"NACGU", dtype=np.uint8)[codes]


def synth_transcript(L, seed):
    return "".join("ACGU"[k] for k in np.random.default_rng(seed).integers(0, 4, L))


def expected_rows(oracle, seq, W, step, r, kind, seed):
    """The TSV rows the reference's loop would write for `seq` with the oracle as its RNA module and the shuffle oracle as its
    scramble(): every window's r+1 energies, structure, centroid, ensemble diversity (one OpenMP thread per window)."""
    starts = scanmod.window_starts(len(seq), W, step)
    rows = ascii_rows(oracle.shuffle_windows(seq, W, step, 0, len(starts), r, kind, seed))
    ref = oracle.scan_windows(rows, len(starts), r)
    return scanmod.rows_from_results(seq, starts, W, r, 37, ref["energies"], ref["structure"], ref["centroid"], ref["ens_div"])


def test_bench_single_rank_over_rccl_cfg2():
    """bench.py --config cfg2 as one rank under torch.distributed.run: init_process_group("nccl"), pack_records on the device,
    all_gather_into_tensor, merge_shards — and 64 windows of what the gather delivered == oracle."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
           "--no-live-counters", "--no-cpu-baseline", "--config", "cfg2"]
    env = dict(os.environ, SCANFOLD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["windows"] == 989 and out["config"]["mfe_folds_per_step"] == 30659
    gc = out["gather_check"]
    assert gc is not None and gc["backend"] == "nccl" and gc["ranks"] == 1 and gc["windows"] == 989
    assert gc["rank0_shard_equals_its_device_tensors"] is True
    assert out["verified_windows"] == 64 and out["verified_mismatches"] == 0 and out["device_status"] == 0
    assert "gathered records" in out["verified_against"]
    assert out["roofline"]["launches"] == 1 and out["roofline"]["folds_per_launch"] == 30659


def test_bench_two_ranks_on_gpu0_over_gloo_cfg2():
    """bench.py with world == 2 on REAL device tensors (the code the driver's multi-GPU run is the first to execute with
    world > 1): both ranks on GPU 0 — RCCL refuses two ranks on one device, so the collective runs over gloo, staged through
    the host for the exchange only (SCANFOLD_DIST_BACKEND=gloo, dist.gather_records) — `pack_records` on the device, one
    all-gather, `merge_shards`; the verified windows come from BOTH shards of the gathered tensor and the elapsed time in the
    line is the maximum over the ranks."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-live-counters", "--no-cpu-baseline", "--config", "cfg2"]
    env = dict(os.environ, SCANFOLD_DIST_BACKEND="gloo", SCANFOLD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SCANFOLD_BENCH_FORCE_DIST", None)
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["windows"] == 989
    gc = out["gather_check"]
    assert gc is not None and gc["backend"] == "gloo" and gc["ranks"] == 2 and gc["windows"] == 989
    assert gc["shard_ranges"] == [[0, 495], [495, 989]]
    assert gc["rank0_shard_equals_its_device_tensors"] is True
    assert out["verified_windows"] == 64 and out["verified_mismatches"] == 0 and out["device_status"] == 0
    per_shard = gc["verified_windows_per_shard"]
    assert len(per_shard) == 2 and min(per_shard) >= 30 and sum(per_shard) == 64
    assert "gathered records" in out["verified_against"]
    assert len(gc["rank_elapsed_s"]) == 2 and gc["elapsed_is_max_over_ranks"] is True
    assert abs(out["ms_per_step"] - max(gc["rank_elapsed_s"]) / 2 * 1e3) < 1e-6
    assert abs(out["value"] - 989 * 2 / max(gc["rank_elapsed_s"])) < 1e-6 * out["value"]
    # rank 0's kernel profile is its own shard's: 495 windows x 31 folds per launch
    assert out["roofline"]["folds_per_launch"] == 495 * 31


def test_all_n_stretch_inside_a_transcript(gpu_engine, oracle):
    """ScanFold-Scan.py:374-380 on the engine: windows that are 120 x N get the literal row, the windows that overlap the
    stretch partly — N folds as a non-pairing nucleotide and is a fifth symbol for the device dinucleotide shuffle — equal
    the oracle-built rows, for both shuffle types."""
    rng = np.random.default_rng(31)
    left = "".join("ACGU"[k] for k in rng.integers(0, 4, 310))
    right = "".join("ACGT"[k] for k in rng.integers(0, 4, 290))  # T: transcribed in the output column
    seq = left + "N" * 150 + right
    W, step, r = 120, 10, 12
    literal = "\t37\t0\t#DIV/0\t0\t0\t" + "N" * 120 + "\t" + "." * 120 + "\t" + "." * 120 + "\n"
    for name, kind in (("di", _lib.SHUFFLE_DI), ("mono", _lib.SHUFFLE_MONO)):
        got = scanmod.scan_record(seq, W, step, r, name, 37, gpu_engine, seed=9)
        exp = expected_rows(oracle, seq, W, step, r, kind, 9)
        assert len(got) == len(scanmod.window_starts(len(seq), W, step)) == 64
        all_n = [k for k, i in enumerate(scanmod.window_starts(len(seq), W, step)) if seq[i:i + W] == "N" * W]
        assert all_n == [31, 32, 33, 34]
        for k in all_n:
            assert got[k] == "%d\t%d" % (k * step + 1, k * step + W) + literal
        assert got == exp
        # the engine's raw output for an all-N window: energy 0 for the native and every shuffle, no pair anywhere
        res = gpu_engine.scan(seq, W, step, 31, 4, r, kind, 9)
        assert (res["energies"] == 0).all() and res["structure"] == ["." * W] * 4 == res["centroid"]
        assert np.abs(res["ens_div"]).max() < 1e-12


@pytest.mark.parametrize("cfg", [(1000, 1, 40, 10, 23), (10000, 2, 10, 30, 989)])
def test_cli_mono_shuffles_cfg1_cfg2_against_the_shuffle_oracle(gpu_engine, oracle, tmp_path, cfg):
    """`-type mono` (the reference's default, ScanFold-Scan.py:43,273-274) through the command line on BASELINE configs 1 and
    2: the file == header + the rows built from the oracle on the shuffle oracle's mononucleotide shuffles, byte for byte."""
    L, tseed, step, r, n_win = cfg
    seq = synth_transcript(L, tseed)
    fa = tmp_path / "in.fa"
    with open(fa, "w") as f:
        f.write(">cfg synthetic\n")
        for k in range(0, len(seq), 70):
            f.write(seq[k:k + 70] + "\n")
    out = tmp_path / "out.tsv"
    assert scanmod.main(["-i", str(fa), "-w", "120", "-s", str(step), "-r", str(r), "-type", "mono", "--seed", "21",
                         "-o", str(out)]) == 0
    exp = expected_rows(oracle, seq, 120, step, r, _lib.SHUFFLE_MONO, 21)
    assert len(exp) == n_win
    assert out.read_text() == scanmod.header_line("cfg") + "".join(exp)
    # and the shuffles behind those rows are permutations of their window that differ from the dinucleotide ones
    mono = gpu_engine.shuffle_windows(seq, 120, step, 0, 3, r, _lib.SHUFFLE_MONO, 21)
    di = gpu_engine.shuffle_windows(seq, 120, step, 0, 3, r, _lib.SHUFFLE_DI, 21)
    assert (np.sort(mono, axis=1) == np.sort(mono[[0] * (r + 1) + [r + 1] * (r + 1) + [2 * (r + 1)] * (r + 1)], axis=1)).all()
    assert (mono != di).any()


def _run_poison(pattern, widths, n):
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from scanfold_amd import _lib\n"
        "eng = _lib.Engine(0)\n"
        "for W in %r:\n"
        "    arr = np.frombuffer(b'ACGU', dtype=np.uint8)[np.random.default_rng(W).integers(0, 4, (%d, W))]\n"
        "    e, db = eng.mfe_trace_batch(arr[:64])\n"
        "    print(W, int(eng.mfe_batch(arr).astype(np.int64).sum()), int(e.sum()), hash(tuple(db)) & 0xffffffff)\n"
        % (ROOT, list(widths), n))
    env = dict(os.environ, PYTHONHASHSEED="0")
    if pattern is None:
        env.pop("SCANFOLD_MFE_POISON", None)
    else:
        env["SCANFOLD_MFE_POISON"] = str(pattern)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    return p.stdout


def test_poisoned_lds_slack_does_not_move_an_energy(gpu_engine, oracle):
    """The straight-line cell code of the short diagonals reads candidates of loop sizes that do not exist yet — up to 23
    words past a rolling row, past the last row into what follows the table — and must be indifferent to what it finds there
    (the size tables charge those sizes 32 767).  SCANFOLD_MFE_POISON=k makes every workgroup refill, before EVERY fold,
    each LDS byte that is not a live table of that fold (the rolling rows no diagonal has written yet, the mirror rows, the
    cell lists, the unused end of the fML area) with an adversarial int16 pattern: -32 768, -28 000, 0, 32 767, or a different
    one of them per entry.  Energies and structures must equal the unpoisoned run at every width class (narrow merged-helper,
    W = 120, generic narrow, wide, W = 200) — and the oracle."""
    widths = (16, 31, 64, 77, 100, 117, 120, 128, 129, 160, 200, 256)
    base = _run_poison(None, widths, 1500)
    for pattern in (1, 2, 3, 4, 5):
        assert _run_poison(pattern, widths, 1500) == base, "pattern %d" % pattern
    # the unpoisoned run is the oracle's (one width per instantiation)
    for W in (77, 120, 200):
        arr = random_seqs(np.random.default_rng(W), 1500, W)
        assert (gpu_engine.mfe_batch(arr) == oracle.mfe_batch(arr)).all()


def test_temperature_path_against_the_independent_python_model(gpu_engine):
    """-t on the HIP engine against tests/py_model.py (a pure-Python restatement of the model in energy space that shares no code
    with the oracle or the library): ensemble energy, ensemble diversity and MFE of enumerable sequences at 25 and 50 C."""
    import py_model
    from par_util import par_text, synthetic_enthalpies
    from scanfold_amd import params
    from test_independent_model import SEQS
    base = params.default_params()
    pset = params.parse_par_text(par_text(base.rec, synthetic_enthalpies(base.rec, 5)), source="synthetic.par")
    try:
        gpu_engine.load_params(pset)
        for T in (25.0, 50.0):
            gpu_engine.set_temperature(T)
            for seq in SEQS:
                dg, dist, count, mfe, db, _ = py_model.ensemble(pset, seq, T)
                o = gpu_engine.pf_batch([seq])
                assert abs(float(o["dG"][0]) - dg) < 1e-9 * max(1.0, abs(dg)), (seq, T, o["dG"][0], dg)
                assert abs(float(o["mean_bp_dist"][0]) - dist) < 1e-9, (seq, T)
                assert int(gpu_engine.mfe_batch([seq])[0]) == mfe, (seq, T)
            from test_independent_model import CONSTRAINED
            for seq, cons in CONSTRAINED:  # -c: the model filters whole structures, the kernels apply the constraint per cell
                dg, dist, count, mfe, db, _ = py_model.ensemble(pset, seq, T, cons=cons)
                o = gpu_engine.fold_constrained([seq], [cons])
                assert abs(float(o["dG"][0]) - dg) < 1e-9 * max(1.0, abs(dg)) and abs(float(o["mean_bp_dist"][0]) - dist) < 1e-9, (seq, cons, T)
                assert int(o["mfe"][0]) == mfe, (seq, cons, T)
    finally:
        gpu_engine.load_params(base)


def test_scratch_slices_with_grids_that_are_not_multiples_of_eight(gpu_engine, oracle):
    """The c + ExtLoop tables (MFE kernel) and the parked partition-function state are placed per XCD — slice
    (b mod 8) * ceil(grid / 8) + b / 8 for workgroup b — so a grid that is not a multiple of eight uses slices beyond `grid`:
    batches of 1 .. 1031 folds (grid = n below the resident 1 024) and scans of 3 .. 37 windows, against the oracle."""
    rng = np.random.default_rng(77)
    for W in (120, 100, 200):
        for n in (1, 7, 9, 63, 1025 if W != 200 else 515, 1031 if W != 200 else 517):
            arr = random_seqs(rng, n, W)
            got = gpu_engine.mfe_batch(arr)
            sel = np.unique(np.concatenate([np.arange(min(n, 8)), np.arange(max(n - 8, 0), n)]))
            assert (got[sel] == oracle.mfe_batch(arr[sel])).all(), (W, n)
            e, db = gpu_engine.mfe_trace_batch(arr[:min(n, 9)])
            for k in range(min(n, 9)):
                odb, oe = oracle.mfe(bytes(arr[k]).decode())
                assert (db[k], e[k]) == (odb, oe), (W, n, k)
    seq = "".join("ACGU"[k] for k in rng.integers(0, 4, 160))
    for n_win in (3, 9, 37):
        res = gpu_engine.scan(seq, 120, 1, 0, n_win, 2, _lib.SHUFFLE_DI, 3)
        for w in (0, n_win // 2, n_win - 1):
            o = oracle.pf(seq[w:w + 120])
            assert abs(o["dG"] - res["ens_dG"][w]) < 1e-9 and o["centroid"] == res["centroid"][w], (n_win, w)
            assert abs(o["mean_bp_dist"] - res["ens_div"][w]) < 1e-9, (n_win, w)

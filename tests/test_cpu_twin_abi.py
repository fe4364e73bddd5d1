"""oracle/libscanfold_cpu.so — the CPU restatement behind the SAME C ABI as libscanfold_hip.so (SURVEY.md §8b "same
symbols from a twin").  Test / baseline infrastructure: the product loads it only when a user points SCANFOLD_LIB_PATH
at it.  Here: every declared symbol is exported; BASELINE config 1 (1 kb, W = 120, step = 40, 10 shuffles — "CPU
reference path, plumbing, no GPU") runs through the unchanged command line of scanfold_amd on top of it and the file
equals the rows rebuilt from the oracle's values; the Fold stage's tabulation equals the numpy grouping bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from scanfold_amd import _lib, fold, params, scan as scanmod
from test_cabi import declared_symbols

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "oracle", "libscanfold_cpu.so")


@pytest.fixture(scope="module")
def cpu_engine():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
    e = _lib.Engine(device=0, lib_path=CPU_LIB)
    yield e
    e.shutdown()


def test_cpu_twin_exports_the_whole_abi(cpu_engine):
    for name in declared_symbols():
        assert getattr(cpu_engine.lib, name) is not None, name
    assert "host CPU" in cpu_engine.device_name()


def test_config1_through_the_command_line_on_the_cpu_twin(tmp_path, oracle):
    seq = "".join("ACGU"[k] for k in np.random.default_rng(1).integers(0, 4, 1000))
    fa = tmp_path / "cfg1.fa"
    fa.write_text(">cfg1 synthetic 1 kb\n" + seq + "\n")
    out = tmp_path / "cfg1.tsv"
    p = subprocess.run([sys.executable, "-m", "scanfold_amd.scan", "-i", str(fa), "-w", "120", "-s", "40", "-r", "10",
                        "-type", "di", "--seed", "7", "-o", str(out)], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, SCANFOLD_LIB_PATH=CPU_LIB, SCANFOLD_DEVICE="0"), timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = out.read_text().split("\n")
    assert lines[0] == scanmod.header_line("cfg1").rstrip("\n") and len(lines) == 1 + 23 + 1
    rows = np.frombuffer(b"NACGU", dtype=np.uint8)[oracle.shuffle_windows(seq, 120, 40, 0, 23, 10, 1, 7)]
    ref = oracle.scan_windows(rows, 23, 10)
    exp = scanmod.rows_from_results(seq, scanmod.window_starts(1000, 120, 40), 120, 10, 37, ref["energies"], ref["structure"],
                                    ref["centroid"], ref["ens_div"])
    assert [ln + "\n" for ln in lines[1:-1]] == exp


def test_cpu_twin_batches_and_tabulation(cpu_engine, oracle):
    cpu_engine.load_params(params.default_params())
    rng = np.random.default_rng(3)
    arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (6, 50))]
    assert (cpu_engine.mfe_batch(arr) == oracle.mfe_batch(arr)).all()
    e, db = cpu_engine.mfe_trace_batch(arr)
    r = cpu_engine.pf_batch(arr)
    for k in range(len(arr)):
        s = bytes(arr[k]).decode()
        o = oracle.pf(s)
        assert (db[k], e[k]) == oracle.mfe(s) and o["centroid"] == r["centroid"][k]
        assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < 1e-9 and abs(o["dG"] - r["dG"][k]) < 1e-9
    # pair tabulation: groups and numpy-order sums equal the host grouping of scanfold_amd.fold
    import json
    c = json.load(open(os.path.join(ROOT, "tests", "golden", "fold_cases.json")))["cases"][1]
    table = fold.ScanTable.from_rows(c["tsv"].split("\n")[2:], "x")
    host = fold.Tabulation(table).groups()
    dev = fold.DeviceTabulation(table, cpu_engine).groups()
    o = np.lexsort((np.asarray(host[3]), np.asarray(host[0])))
    for a, b in zip(host, dev):
        assert np.array_equal(np.asarray(a)[o], np.asarray(b))
    from scanfold_amd._lib import ScanFoldHipError
    with pytest.raises(ScanFoldHipError, match="scan table"):
        cpu_engine.tabulate_pairs(["((..", "...."], [1, 2], np.zeros(2), np.zeros(2), np.zeros(2))

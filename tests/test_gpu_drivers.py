"""The widened drivers on the real GPU through the C ABI: the Fold stage with the device pair
tabulation on the reference-written cases, the ScanFold.py flavour of the scan (plain, --constraints, --react) against
rows rebuilt from oracle values, and the whole scanfold.main pipeline once.  The CPU suite runs the same code on the
emulated engine (tests/test_fold.py, tests/test_scanfold_driver.py); here libscanfold_hip.so does the folds."""
import hashlib
import json
import os

import numpy as np
import pytest

from scanfold_amd import RNA, _lib, fold, functions as sff, motifs
from scanfold_amd import scanfold as sfd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _f32(dcal):
    return float(np.float32(dcal) / np.float32(100))


def test_fold_stage_on_gpu_reproduces_every_reference_file(gpu_engine, tmp_path):
    """fold.fold(table, ..., engine=<HIP>): pair tabulation by sf_tabulate_pairs on the MI355X, all four cases of
    tests/golden/fold_cases.json (files ScanFold-Fold.py wrote: ScanFold-Fold.py:466-1074) — every file byte-identical,
    the CT files too once their first line (it carries the output path) is set aside."""
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "fold_cases.json")))["cases"]
    assert len(cases) == 4
    for c in cases:
        lines = c["tsv"].split("\n")
        table = fold.ScanTable.from_rows(lines[2:], lines[0].split("\t")[-1].strip())
        d = tmp_path / ("gpu%d" % c["seed"])
        d.mkdir()
        fold.fold(table, str(d / "scan.tsv.ScanFold."), bp_path=str(d / "final_partners_test.bp"), engine=gpu_engine)
        assert sorted(os.listdir(d)) == sorted(c["outputs"]), c["seed"]
        for name, exp in c["outputs"].items():
            text = (d / name).read_text()
            if name.endswith(".ct"):
                assert isinstance(exp, str) and text.split("\n", 1)[1] == exp.split("\n", 1)[1], (c["seed"], name)
            elif isinstance(exp, dict):
                assert hashlib.sha256(text.encode()).hexdigest() == exp["sha256"], (c["seed"], name)
            else:
                assert text == exp, (c["seed"], name)


def _record(n, seed):
    rng = np.random.default_rng(seed)
    stem = "GGCGCGGCACGCAGG"
    comp = stem[::-1].translate(str.maketrans("ACGU", "UGCA"))
    body = "".join("ACGU"[k] for k in rng.integers(0, 4, n - 2 * len(stem) - 5))
    return body[:200] + stem + "GAAAC" + comp + body[200:]


def test_scanfold_py_rows_plain_constraints_react_on_gpu(gpu_engine, oracle):
    """scanfold.scan_rows (ScanFold.py:420-757) on a 600-nt record, W = 120, step = 20, r = 12: every row of the three
    flavours against values rebuilt from the oracle — shuffles by the shuffle oracle, SFF z-score / p-value, native fold
    plain / under the window's constraint slice / under the Deigan term (and then the centroid of the UNconstrained
    partition function, ScanFold.py:522-544)."""
    seq = _record(600, 21)
    W, step, r, seed = 120, 20, 12, 5
    n_win = (len(seq) - W) // step + 1
    shuf = np.frombuffer(b"NACGU", dtype=np.uint8)[oracle.shuffle_windows(seq, W, step, 0, n_win, r, 1, seed)]
    E = oracle.mfe_batch(shuf).reshape(n_win, r + 1)
    cons = "".join("x" if k % 17 == 3 else "." for k in range(len(seq)))
    rng = np.random.default_rng(9)
    react = [-999.0] + [float(x) for x in np.round(rng.uniform(0, 2, len(seq)), 2)]
    for k in range(40, len(react), 53):
        react[k] = -999.0
    flavours = {
        "plain": dict(),
        "constraints": dict(constraints=cons),
        "react": dict(reactivities=react, slope=1.8, intercept=-0.6),
    }
    try:
        for name, kw in flavours.items():
            rows, table = sfd.scan_rows(seq, W, step, r, "di", 37, gpu_engine, seed=seed, **kw)
            assert len(rows) == n_win == len(table.starts)
            for k, row in enumerate(rows):
                f = row.rstrip("\n").split("\t")
                i = step * k
                frag = seq[i:i + W]
                el = [_f32(v) for v in E[k]]
                assert f[0] == str(i + 1) and f[1] == str(i + W) and f[7] == frag and f[10] == str(sff.get_gc_content(frag))
                assert f[4] == str(round(sff.zscore_function(el, r), 2)) and f[5] == str(round(sff.pvalue_function(el, r), 2))
                if name == "constraints":
                    oracle.set_constraint(cons[i:i + W], None)
                elif name == "react":
                    oracle.set_constraint(None, RNA.deigan_pseudo_energies(react[i + 1:i + W + 1], 1.8, -0.6, W))
                db, e = oracle.mfe(frag)
                o = oracle.pf(frag) if name != "react" else None
                oracle.set_constraint(None, None)
                if o is None:
                    o = oracle.pf(frag)
                assert (f[8], f[3]) == (db, str(round(_f32(e), 2))), (name, k)
                assert f[9] == o["centroid"] and f[6] == str(round(o["mean_bp_dist"], 2)), (name, k)
    finally:
        oracle.set_constraint(None, None)


def test_scanfold_main_pipeline_on_gpu(gpu_engine, oracle, tmp_path, monkeypatch):
    """scanfold.main once, end to end (ScanFold.py:420-757,1036-1500,1582-1776): scan -> .out table -> Fold stage (device
    tabulation) -> CT / bp / dbn / wig / fasta files -> motif extraction with constrained refolds."""
    monkeypatch.setattr(_lib, "_engine", gpu_engine)
    monkeypatch.chdir(tmp_path)
    seq = _record(420, 33).replace("U", "T")
    (tmp_path / "in.fa").write_text(">rec9 x\n" + seq + "\n")
    assert sfd.main(["in.fa", "-w", "120", "-s", "10", "-r", "8", "--type", "di", "--seed", "3"]) == 0
    base = "rec9.win_120.stp_10.rnd_8.shfl_di"
    lines = (tmp_path / (base + ".out")).read_text().split("\n")
    assert lines[0] == sfd.header_line("rec9").rstrip("\n")
    tseq = seq.replace("T", "U")
    rows = [ln.split("\t") for ln in lines[1:-1]]
    n_win = (len(seq) - 120) // 10 + 1
    assert len(rows) == n_win and all(len(r) == 11 for r in rows)
    shuf = np.frombuffer(b"NACGU", dtype=np.uint8)[oracle.shuffle_windows(tseq, 120, 10, 0, n_win, 8, 1, 3)]
    E = oracle.mfe_batch(shuf).reshape(n_win, 9)
    for k, r in enumerate(rows):
        frag = tseq[10 * k:10 * k + 120]
        el = [_f32(v) for v in E[k]]
        db, e = oracle.mfe(frag)
        assert r[7] == frag and r[8] == db and r[3] == str(round(el[0], 2))
        assert r[4] == str(round(sff.zscore_function(el, 8), 2)) and r[5] == str(round(sff.pvalue_function(el, 8), 2))
    table = fold.ScanTable("rec9", [int(r[0]) for r in rows], [float(r[3]) for r in rows], [float(r[4]) for r in rows],
                           [float(r[6]) for r in rows], [r[7] for r in rows], [r[8] for r in rows])
    tab = fold.Tabulation(table)
    res = fold.compete(tab, fold.best_partners(tab))
    dbn = (tmp_path / (base + ".ScanFold.-2.dbn")).read_text().split("\n")
    assert dbn[0] == ">Zavg_-2" and dbn[1] == tseq and dbn[2] == fold.structure_string(tab, res, -2.0)
    ct = (tmp_path / (base + ".ScanFold.-1.ct")).read_text().split("\n")
    assert int(ct[0].split("\t")[0]) == len(seq)
    assert sum(1 for ln in ct[1:-1] if int(ln.split()[4]) != 0) == 2 * fold.structure_string(tab, res, -1.0).count("(")
    assert os.path.exists(tmp_path / (base + ".ScanFold.final_partners.txt")) and os.path.exists(tmp_path / (base + ".bp"))
    gff = (tmp_path / (base + ".ExtractedStructures.gff3")).read_text().split("\n")[:-1]
    ms = motifs.extract_structures(dbn[2] + "\n", tseq, verbose=False)
    assert len(gff) == len(ms)
    try:
        for ln, m in zip(gff, ms):
            att = dict(kv.split("=", 1) for kv in ln.split("\t")[8].split(";")[1:])
            oracle.set_constraint(m.structure, None)
            db, e = oracle.mfe(m.sequence)
            assert att["refoldedMFE"] == db and att["MFE(kcal/mol)"] == str(round(_f32(e), 2))
    finally:
        oracle.set_constraint(None, None)

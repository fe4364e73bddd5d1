"""The Fold stage against files the REFERENCE wrote: tests/golden/fold_cases.json holds Scan TSVs and
everything `python /root/reference/ScanFold-Fold.py -i scan.tsv` produced from them (made by
tests/golden/make_golden_fold.py in the build container).  scanfold_amd.fold must reproduce every file byte for byte:
the CT files of all filters, the IGV .bp track, the final-partners log and the per-nucleotide log."""
import hashlib
import json
import os

import numpy as np
import pytest

from scanfold_amd import fold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(ROOT, "tests", "golden", "fold_cases.json")) as f:
        return json.load(f)["cases"]


def test_fold_outputs_are_byte_identical_to_the_reference(cases, tmp_path, monkeypatch):
    for c in cases:
        d = tmp_path / ("case%d" % c["seed"])
        d.mkdir()
        (d / "scan.tsv").write_text(c["tsv"])
        monkeypatch.chdir(d)  # the reference names its files relative to the working directory
        assert fold.main(["-i", "scan.tsv"]) == 0
        got = sorted(n for n in os.listdir(d) if n != "scan.tsv")
        assert got == sorted(c["outputs"]), (c["seed"], got)
        for name, exp in c["outputs"].items():
            text = (d / name).read_text()
            if isinstance(exp, dict):
                assert text.count("\n") == exp["lines"] and "".join(text.splitlines(True)[:60]) == exp["head"], name
                assert hashlib.sha256(text.encode()).hexdigest() == exp["sha256"], name
            else:
                assert text == exp, (c["seed"], name)


def test_group_sums_are_numpy_sums_bit_for_bit():
    rng = np.random.default_rng(1)
    vals = np.round(rng.normal(0, 2, 120000), 2)
    cnt = rng.integers(1, 300, 1500)
    cnt = cnt[np.cumsum(cnt) <= len(vals)]
    st = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    got = fold.group_sums(vals, st, cnt)
    exp = np.array([np.sum(list(vals[s:s + c])) for s, c in zip(st, cnt)])
    assert np.array_equal(got, exp)
    assert fold.group_sums(np.array([-0.1, -0.1, -0.1]), np.array([0]), np.array([3]))[0] == np.sum([-0.1, -0.1, -0.1])


def test_pair_partners_and_consensus_structure(cases):
    S = np.frombuffer(b"((..))((..))" + b"(.(...).)..." + b"............", dtype=np.uint8).reshape(3, 12)
    p = fold.pair_partners(S)
    assert p[0].tolist() == [5, 4, -1, -1, 1, 0, 11, 10, -1, -1, 7, 6]
    assert p[1].tolist() == [8, -1, 6, -1, -1, -1, 2, -1, 0, -1, -1, -1] and (p[2] == -1).all()
    with pytest.raises(ValueError):
        fold.pair_partners(np.frombuffer(b"((..", dtype=np.uint8).reshape(1, 4))
    # the planted hairpin of case 1 comes out of the -1 filter as a helix
    c = cases[0]
    table = fold.ScanTable.from_rows(c["tsv"].split("\n")[2:], "x")  # header and the row the reference drops
    tab = fold.Tabulation(table)
    res = fold.compete(tab, fold.best_partners(tab))
    db = fold.structure_string(tab, res, -1.0)
    assert db.count("(") == db.count(")") >= 8
    ct = c["outputs"]["scan.tsv.ScanFold.-1.ct"].split("\n")[1:-1]
    pairs_ct = sum(1 for ln in ct if int(ln.split()[4]) != 0)
    assert pairs_ct == 2 * db.count("(")

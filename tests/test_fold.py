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


def test_competition_allowed_mode_writes_the_reference_dp_files(cases, tmp_path, monkeypatch):
    """`-c 0` (ScanFold-Fold.py:1022-1038; write_dp :380-394): DP files of the best partners for the five filters,
    best_bps_test.bp, a final-partners log with its header only — byte for byte what the reference wrote."""
    for c in cases:
        d = tmp_path / ("c0_%d" % c["seed"])
        d.mkdir()
        (d / "scan.tsv").write_text(c["tsv"])
        monkeypatch.chdir(d)
        assert fold.main(["-i", "scan.tsv", "-c", "0"]) == 0
        got = sorted(n for n in os.listdir(d) if n != "scan.tsv" and not n.endswith(".log.txt"))
        assert got == sorted(c["outputs_c0"]), (c["seed"], got)
        for name, exp in c["outputs_c0"].items():
            assert (d / name).read_text() == exp, (c["seed"], name)
        assert len([n for n in got if n.endswith(".dp")]) == 5


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


# ---- the device tabulation (sf_tabulate_pairs), kernel source compiled for the CPU (tests/emul) ----
@pytest.fixture(scope="module")
def emul():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    from emul_engine import emul_engine
    e = emul_engine()
    yield e
    e.shutdown()


def _sorted_groups(groups):
    gk, gj, gcount, gfirst = (np.asarray(a, dtype=np.int64) for a in groups[:4])
    o = np.lexsort((gfirst, gk))
    return [a[o] for a in (gk, gj, gcount, gfirst)] + [np.asarray(a)[o] for a in groups[4:]]


def _random_structure(rng, W):
    s = ["."] * W
    stack_room = W
    p = 0
    while p < W - 8:
        if rng.random() < 0.35:
            ln, loop = int(rng.integers(2, 6)), int(rng.integers(3, 9))
            if p + 2 * ln + loop < W:
                for t in range(ln):
                    s[p + t], s[p + 2 * ln + loop - 1 - t] = "(", ")"
                p += 2 * ln + loop
                continue
        p += int(rng.integers(1, 5))
    return "".join(s)


def _rebalance(st):
    out, stack = list(st), []
    for p, ch in enumerate(st):
        if ch == "(":
            stack.append(p)
        elif ch == ")":
            if stack:
                stack.pop()
            else:
                out[p] = "."
    for p in stack:
        out[p] = "."
    return "".join(out)


def _synthetic_table(rng, n, W, step, start=1, n_structs=5, open_prob=None):
    structs = [_random_structure(rng, W) for _ in range(n_structs)]
    if open_prob is not None:  # mostly unpaired windows: long (k, k) groups
        structs = ["".join(ch if rng.random() < open_prob else "." for ch in st) for st in structs]
        structs = [st if st.count("(") == st.count(")") and fold.pair_partners(np.frombuffer(st.encode(), dtype=np.uint8)[None, :]) is not None
                   else "." * W for st in map(_rebalance, structs)]
    pick = rng.integers(0, n_structs, n)
    z = np.round(rng.normal(-0.4, 1.5, n), 2)
    mfe = np.round(rng.normal(-20, 6, n), 1)
    ed = np.round(np.abs(rng.normal(20, 8, n)), 2)
    seq = "".join("ACGU"[v] for v in rng.integers(0, 4, start + n * step + W))
    starts = start + step * np.arange(n)
    return fold.ScanTable("syn", starts, mfe, z, ed, [seq[s:s + W] for s in starts], [structs[v] for v in pick])


def test_device_tabulation_equals_host_grouping_bit_for_bit(emul, cases):
    rng = np.random.default_rng(8)
    tables = [fold.ScanTable.from_rows(c["tsv"].split("\n")[2:], "x") for c in cases]
    tables += [_synthetic_table(rng, 90, 40, 1), _synthetic_table(rng, 40, 33, 7, start=5), _synthetic_table(rng, 9, 20, 25),
               _synthetic_table(rng, 330, 200, 1, n_structs=3, open_prob=0.1),   # groups of up to 200 windows: the pairwise split above 128
               _synthetic_table(rng, 420, 400, 1, n_structs=2, open_prob=0.05), _synthetic_table(rng, 1, 16, 1)]
    for t in tables:
        host = _sorted_groups(fold.Tabulation(t).groups())
        dev = fold.DeviceTabulation(t, emul).groups()
        assert np.array_equal(dev[0][1:] >= dev[0][:-1], np.ones(len(dev[0]) - 1, dtype=bool))  # ordered by nucleotide
        devs = _sorted_groups(dev)
        for a, b in zip(host, devs):
            assert a.dtype == b.dtype and np.array_equal(a, b)
        # ... and already in (nucleotide, first window) order as they leave the device
        for a, b in zip(dev, devs):
            assert np.array_equal(np.asarray(a), b)
    assert int(fold.Tabulation(tables[-3]).groups()[2].max()) > 128 and int(fold.Tabulation(tables[-2]).groups()[2].max()) > 256


def test_fold_through_the_device_tabulation_is_byte_identical(emul, cases, tmp_path):
    for c in cases[:2]:
        table = fold.ScanTable.from_rows(c["tsv"].split("\n")[2:], c["tsv"].split("\n")[0].split("\t")[-1].strip())
        d = tmp_path / ("dev%d" % c["seed"])
        d.mkdir()
        fold.fold(table, str(d / "scan.tsv.ScanFold."), bp_path=str(d / "final_partners_test.bp"), engine=emul)
        for name, exp in c["outputs"].items():
            if name.endswith(".ct"):
                continue  # the CT header carries the path
            text = (d / name).read_text()
            if isinstance(exp, dict):
                assert hashlib.sha256(text.encode()).hexdigest() == exp["sha256"], name
            else:
                assert text == exp, (c["seed"], name)


def test_device_tabulation_rejects_bad_tables(emul):
    from scanfold_amd._lib import ScanFoldHipError
    z = np.zeros(2)
    with pytest.raises(ScanFoldHipError, match="scan table"):
        emul.tabulate_pairs(["((..", "...."], [1, 2], z, z, z)
    with pytest.raises(ScanFoldHipError, match="scan table"):
        emul.tabulate_pairs(["(..)", "...."], [2, 2], z, z, z)
    # rows of nothing but '(' are deeper than the bracket stack (W/2 + 1 entries per window): rejected, nothing
    # written out of bounds (the emulation build aborts on heap corruption; the sanitizer build sees the store itself)
    for W in (30, 400):
        with pytest.raises(ScanFoldHipError, match="scan table"):
            emul.tabulate_pairs(["(" * W, "(" * (W // 2 + 2) + "." * (W - W // 2 - 2)], [1, 2], z, z, z)
    g = emul.tabulate_pairs(["(..)", "(..)"], [1, 2], np.array([-1.0, -2.0]), np.array([-3.0, -4.0]), np.array([1.0, 2.0]))
    assert g["k"].tolist() == [1, 2, 2, 3, 4, 4, 5] and g["j"].tolist() == [4, 2, 5, 3, 1, 4, 2]
    assert g["windows"].tolist() == [1, 1, 1, 2, 1, 1, 1] and g["first_window"].tolist() == [0, 0, 1, 0, 0, 1, 1]
    assert g["sum_z"].tolist() == [-1.0, -1.0, -2.0, -3.0, -1.0, -2.0, -2.0] and g["sum_ed"][3] == 3.0

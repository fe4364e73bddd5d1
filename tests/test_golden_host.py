"""Host logic vs golden vectors captured from the reference's pure-Python half (tests/golden/make_golden.py)."""
import math
import random
import statistics

import numpy as np
import pytest

from scanfold_amd import functions as sff
from scanfold_amd import scan as scanmod
from scanfold_amd import scan_functions as sfn


def test_dinuclshuffle_matches_reference_strings_and_rng_consumption(golden):
    for item in golden["dinucl"]:
        random.seed(item["seed"])
        out = sff.dinuclShuffle(item["s"])
        assert out == item["out"], item["seed"]
        # same number of random draws as the reference: the generator state afterwards is identical
        assert list(random.getstate()[1][:3]) == item["state_after"]


def test_dinuclshuffle_invariants():
    rng = random.Random(1)
    for _ in range(200):
        s = "".join(rng.choice("ACGU") for _ in range(rng.randint(8, 150)))
        t = sff.dinuclShuffle(s)
        assert len(t) == len(s) and t[0] == s[0] and t[-1] == s[-1]
        assert sff.get_dinucleotide_counts(t) == sff.get_dinucleotide_counts(s)


def test_dinuclshuffle_rejects_non_acgu_like_reference():
    with pytest.raises(KeyError):
        sff.dinuclShuffle("GGNAAC")


def test_scramble_and_randomizer(golden):
    for item in golden["scramble_di"]:
        random.seed(item["seed"])
        assert sff.scramble(item["text"], item["r"], "di") == item["out_sff"]
        random.seed(item["seed"])
        assert sfn.scramble(item["text"].replace("T", "U"), item["r"], "di") == item["out_scan_on_transcribed"]
    for item in golden["randomizer"]:
        random.seed(item["seed"])
        assert sff.randomizer(item["frag"]) == item["out"]
    assert sff.scramble("ACGU", 3, "bogus") == []
    mono = sff.scramble("AACCGGUU", 4, "mono")
    assert len(mono) == 4 and all(sorted(m) == sorted("AACCGGUU") for m in mono)


def test_zscore_pscore_both_variants(golden):
    for item in golden["zscore"]:
        E, r = item["energy_list"], item["r"]
        if "sff_error" in item:
            with pytest.raises(statistics.StatisticsError):
                sff.zscore_function(E, r)
        else:
            assert sff.zscore_function(E, r) == item["sff"]
        z = sfn.zscore_function(E, r)
        if isinstance(item["scan"], str):
            assert z == item["scan"]
        elif math.isnan(item["scan"]):
            assert math.isnan(z)
        else:
            assert float(z) == item["scan"]
        assert sff.pvalue_function(E, r) == item["pvalue"]
        assert sfn.pscore_function(E, r) == item["pscore"]
        assert str(round(sfn.pscore_function(E, r), 2)) == item["pscore_rounded_str"]


def test_rowwise_statistics_are_bit_equal_to_per_row_calls(golden):
    rng = np.random.default_rng(0)
    for r in (2, 5, 10, 30, 100):
        E = (rng.integers(-4000, -500, (300, r + 1)).astype(np.float32) / np.float32(100)).astype(np.float64)
        E[5] = E[5, 0]
        z, sd0 = sfn.zscores_rows(E, r)
        p = sfn.pscores_rows(E)
        for k in range(len(E)):
            row = [float(v) for v in E[k]]
            zr = sfn.zscore_function(row, r)
            if isinstance(zr, str):
                assert sd0[k]
            else:
                assert not sd0[k] and z[k] == zr and str(round(np.float64(z[k]), 2)) == str(round(zr, 2))
            assert p[k] == sfn.pscore_function(row, r)


def test_helpers(golden):
    for item in golden["helpers"]:
        assert sff.get_gc_content(item["frag"]) == item["gc"]
        assert sff.simple_transcribe(item["frag"]) == item["transcribe"]
        assert sff.get_dinucleotide_counts(item["frag"]) == item["dicounts"]


def test_tsv_rows_and_header_byte_for_byte(golden):
    for item in golden["tsv"]:
        assert scanmod.header_line(item["read_name"]) == item["header"]
        row = scanmod.format_row(item["start_nucleotide"], item["end_nucleotide"], item["temperature"], item["MFE"],
                                 item["zscore"], item["pscore"], item["ED"], item["frag"], item["structure"],
                                 item["centroid"])
        assert row == item["row"]


def test_rows_from_results_reproduces_reference_rounding(golden):
    # feed the golden energy lists through the driver's vectorised path and compare whole rows
    for item in golden["tsv"]:
        if item["zscore"] == "#DIV/0":
            continue  # the literal all-N shortcut row, covered below
        W = len(item["frag"])
        dcal = np.array([[int(round(e * 100)) for e in item["energy_list"]]], dtype=np.int32)
        seq = "A" * (item["start_nucleotide"] - 1) + item["frag"]
        rows = scanmod.rows_from_results(seq, [item["start_nucleotide"] - 1], W, 10, 37, dcal, [item["structure"]],
                                         [item["centroid"]], np.array([item["ED"]]))
        assert rows[0] == item["row"]


def test_all_n_window_shortcut():
    seq = "N" * 120
    rows = scanmod.rows_from_results(seq, [0], 120, 3, 37, np.zeros((1, 4), dtype=np.int32), ["x"], ["y"],
                                     np.zeros(1))
    assert rows[0] == "1\t120\t37\t0\t#DIV/0\t0\t0\t" + "N" * 120 + "\t" + "." * 120 + "\t" + "." * 120 + "\n"


def test_window_enumeration(golden):
    for item in golden["windows"]:
        starts = scanmod.window_starts(item["L"], item["W"], item["step"])
        assert len(starts) == item["n"] == (item["L"] - item["W"]) // item["step"] + 1
        assert starts[:3] == item["first"] and starts[-1] == item["last"]


def test_fasta_reader(tmp_path):
    p = tmp_path / "x.fa"
    p.write_text(">rec1 some description\nACGT\nacgu\n\n>rec2\nGG GG\n")
    assert scanmod.read_fasta(str(p)) == [("rec1", "ACGTacgu"), ("rec2", "GGGG")]
    assert scanmod.transcribe("ACGTacgt") == "ACGUacgu"


def test_vectorised_rows_equal_the_per_row_reference_expressions():
    """rows_from_results builds whole columns at once; every row must still be what the reference's per-window
    expressions give (ScanFold-Scan.py:386,389,426-433,442, restated here with the per-row functions that
    tests above pin to the reference): 4 000 windows incl. sd == 0, ties, halfway values, uint8-array text input."""
    rng = np.random.default_rng(12)
    n, W, r = 4000, 30, 20
    seq = "".join("ACGT"[k] for k in rng.integers(0, 4, n + W - 1))
    dcal = rng.integers(-3000, 100, (n, r + 1)).astype(np.int32)
    dcal[5] = -1230                      # sd == 0
    dcal[6, 1:] = dcal[6, 0]             # sd == 0 again
    dcal[7, 1:] = dcal[7, 0] + 5         # every shuffle above the native: p == 0
    dcal[8] = np.arange(r + 1) * 5 - 25  # regular spacing: halfway-ish z values
    ed = np.round(rng.uniform(0, 40, n), 3)
    ed[:50] = np.arange(50) / 8.0 + 0.005
    db = np.frombuffer(b".()", dtype=np.uint8)[rng.integers(0, 3, (n, W + 1))]
    cen = np.frombuffer(b".()", dtype=np.uint8)[rng.integers(0, 3, (n, W + 1))]
    starts = list(range(n))
    got = scanmod.rows_from_results(seq, starts, W, r, 37, dcal, db, cen, ed)
    E = scanmod.dcal_to_float(dcal)
    for k in list(range(60)) + list(range(60, n, 37)):
        el = [float(v) for v in E[k]]
        MFE = round(el[0], 2)
        try:
            zscore = round(sfn.zscore_function(el, r), 2)
        except Exception:
            zscore = sfn.zscore_function(el, r)
        pscore = round(sfn.pscore_function(el, r), 2)
        ED = round(float(ed[k]), 2)
        frag = seq[k:k + W].replace("T", "U")
        exp = scanmod.format_row(k + 1, k + W, 37, MFE, zscore, pscore, ED, frag, bytes(db[k, :W]).decode(),
                                 bytes(cen[k, :W]).decode())
        assert got[k] == exp, k
    assert got[5].split("\t")[4] == "#DIV/0!"

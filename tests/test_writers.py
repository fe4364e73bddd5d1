"""scanfold_amd.writers against the reference's own writer functions (tests/golden/writers.json)."""
import json
import os

from scanfold_amd import writers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "writers.json")))


def test_wig(tmp_path):
    for it in G["wig"]:
        p = tmp_path / "w.wig"
        writers.write_wig(it["metrics"], it["step"], it["name"], str(p))
        assert p.read_text() == it["out"]


def test_wig_dict(tmp_path):
    for it in G["wig_dict"]:
        p = tmp_path / "wd.wig"
        writers.write_wig_dict(it["zscores"], str(p), it["name"], it["step"])
        assert p.read_text() == it["out"]


def test_fasta_and_fai(tmp_path):
    for it in G["fasta"]:
        pf, pi = tmp_path / "x.fa", tmp_path / "x.fai"
        writers.write_fasta(it["seq"], str(pf), it["name"])
        writers.write_fai(len(it["seq"]), str(pi), it["name"])
        assert pf.read_text() == it["fasta"] and pi.read_text() == it["fai"]


def test_makedbn_incl_a_crossing_pair(tmp_path):
    for k, it in enumerate(G["dbn"]):
        base = tmp_path / ("c%d" % k)
        (tmp_path / ("c%d.ct" % k)).write_text(it["ct"])
        writers.makedbn(str(base), it["name"])
        assert (tmp_path / ("c%d.dbn" % k)).read_text() == it["dbn"], k
    assert "<" in G["dbn"][2]["dbn"] or ">" in G["dbn"][2]["dbn"]

#!/usr/bin/env python3
"""Generate tests/golden/writers.json by RUNNING the reference's writer functions (ScanFoldFunctions imported with a
stub RNA module) on small inputs: write_wig, write_fasta, write_fai, makedbn.  CT inputs: two of the CT files the
reference's Fold stage wrote for tests/golden/fold_cases.json plus a hand-made CT with a crossing pair.
Run in the build container only:   python tests/golden/make_golden_writers.py"""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "writers.json")


class Nuc:
    def __init__(self, n):
        self.nucleotide = n


class Pair:
    def __init__(self, z):
        self.zscore = z


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("RNA", types.ModuleType("RNA"))
    sys.path.insert(0, "/root/reference")
    import ScanFoldFunctions as sff
    cases = json.load(open(os.path.join(HERE, "fold_cases.json")))["cases"]
    G = {"generated_by": "tests/golden/make_golden_writers.py", "wig": [], "fasta": [], "dbn": [], "wig_dict": [], "dp": []}
    with tempfile.TemporaryDirectory() as d:
        for k, (metrics, step) in enumerate([([-1.25, 0.5, "#DIV/0!", 3, -0.0049999, 12.3456789], 1), ([0.1] * 3, 10)]):
            p = os.path.join(d, "w%d.wig" % k)
            sff.write_wig(metrics, step, "chr%d" % k, p)
            G["wig"].append({"metrics": metrics, "step": step, "name": "chr%d" % k, "out": open(p).read()})
        # write_wig_dict (ScanFoldFunctions.py:616-624): one mean z-score per nucleotide of the final partners
        for k, (zs, step) in enumerate([([-2.5, -0.333333333, 0.0, 1.0, -1e-7, 12.3456789], 1), ([0.25, -3.0], 5)]):
            p = os.path.join(d, "wd%d.wig" % k)
            sff.write_wig_dict({i + 1: Pair(z) for i, z in enumerate(zs)}, p, "rec%d" % k, step)
            G["wig_dict"].append({"zscores": zs, "step": step, "name": "rec%d" % k, "out": open(p).read()})
        for seq, name in (("ACGUACGGGAUC", "rec1"), ("G" * 70, "a longer name|with|bars")):
            nd = {i + 1: Nuc(ch) for i, ch in enumerate(seq)}
            pf, pi = os.path.join(d, "x.fa"), os.path.join(d, "x.fai")
            sff.write_fasta(nd, pf, name)
            sff.write_fai(nd, pi, name)
            G["fasta"].append({"seq": seq, "name": name, "fasta": open(pf).read(), "fai": open(pi).read()})
        cts = [cases[0]["outputs"]["scan.tsv.ScanFold.-1.ct"], cases[3]["outputs"]["scan.tsv.ScanFold.no_filter.ct"]]
        cross = ["12\tcrossing"] + ["%d %s %d %d %d %d" % (i, "ACGUACGUACGU"[i - 1], i - 1, i + 1, p, i)
                                    for i, p in zip(range(1, 13), [8, 0, 10, 0, 0, 0, 0, 1, 0, 3, 0, 0])]
        cts.append("\n".join(cross) + "\n")
        for k, text in enumerate(cts):
            base = os.path.join(d, "c%d" % k)
            open(base + ".ct", "w").write(text)
            sff.makedbn(base, "name%d" % k)
            G["dbn"].append({"ct": text, "name": "name%d" % k, "dbn": open(base + ".dbn").read()})
    json.dump(G, open(OUT, "w"), indent=0)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()

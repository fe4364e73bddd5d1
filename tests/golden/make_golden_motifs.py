#!/usr/bin/env python3
"""Generate tests/golden/motifs.json by RUNNING the reference's structure-extraction block and motif refold loop.

ScanFold.py is a script, not a module: the two blocks (located by their first / last statements, never copied) are
read from /root/reference at generation time, dedented and exec'd in a namespace that holds what the script has at
that point — the makedbn line, the record's sequence, the 1-based nucleotide table — with ScanFoldFunctions imported
for NucStructure / ExtractedStructure / dbn2ct / zscore_function / pvalue_function.  ViennaRNA is absent, so the refold
loop runs against a canned RNA module and canned `scramble` / `energies`: the fixture pins the bookkeeping and every
output byte (gff3 line, motif dbn, motif ct) given those fold results, not the folds themselves (those are the
engine's, checked against the oracle in tests/test_gpu_parity.py).
Run in the build container only:   python tests/golden/make_golden_motifs.py"""
import contextlib
import io
import json
import os
import sys
import tempfile
import textwrap
import types

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "motifs.json")
REF = "/root/reference/ScanFold.py"

CASES = [
    # (sequence, structure line with its newline)
    ("GGGAAACCCAUGGGCAAAAGCCCAUUUGCGCAAAAGCGCA", ".((...))...(((.....)))...((((....))))..\n"),
    ("GGGAAACCCAUGGGCAAAAGCCCAUUUGCGCAAAAGCGCA", "(((...)))..(((.....)))...((((....))))..\n"),       # a start at index 0
    ("ACGUACGUACGUACGUACGUACGUACGUACGUACGUACGU", "..((..((...))..((...))..))....<<..>>....\n"),      # multiloop + crossing marks
    ("ACGUACGUACGUACGUACGUACGUACGUACGUACGUACGU", "..((..<<...))..>>...((..{{..))..}}......\n"),      # '<' at depth 1, '>' at depth 0
    ("ACGUACGUACGUACGUACGU", "....................\n"),
    ("ACGUACGUACGUACGUACGU", "..((....))..((...))."),                                                     # no newline: last char dropped
    ("ACGUACGUACGUACGUACGU", "..((..x.))..((...)).\n"),                                                   # an unknown character
]


def block(lines, first, last_exclusive):
    a = next(k for k, ln in enumerate(lines) if ln.strip() == first)
    b = next(k for k, ln in enumerate(lines) if k > a and ln.strip() == last_exclusive)
    return textwrap.dedent("".join(lines[a:b]))


def canned_fold(frag, constraint):
    """Deterministic stand-in fold results: structure = the constraint's brackets, numbers from the lengths."""
    st = "".join(ch if ch in "()" else "." for ch in constraint)
    depth, ok = 0, True
    for ch in st:
        depth += ch == "("
        depth -= ch == ")"
        ok &= depth >= 0
    if not ok or depth:
        st = "." * len(st)
    mfe = -(len(frag) * 37 % 1000) / 100.0 - 0.004999
    ed = (len(frag) * 13 % 700) / 100.0 + 0.005001
    return st, mfe, ed


def canned_energies(seqlist):
    base = -(len(seqlist[0]) % 17) - 3.0
    return [base] + [round(base + 0.1 * ((k * 7) % 23) - 0.6, 2) for k in range(1, len(seqlist))]


def main():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("RNA", types.ModuleType("RNA"))
    sys.path.insert(0, "/root/reference")
    import ScanFoldFunctions as sff
    lines = open(REF).read().splitlines(keepends=True)
    extract_src = block(lines, "structure_raw = filter2constraints", "zscore_total = []")
    x0 = next(k for k, ln in enumerate(lines) if ln.strip() == "structure_raw = filter2constraints")
    a = next(k for k, ln in enumerate(lines) if k > x0 and ln.strip() == "zscore_total = []")  # the result lists come first
    b = next(k for k, ln in enumerate(lines) if k > a and ln.strip() == "se.close()")
    refold_src = textwrap.dedent("".join(lines[a:b + 1]))

    class FC:
        def __init__(self, frag, md=None):
            self.frag, self.cons = frag, "." * len(frag)

        def hc_add_from_db(self, c):
            self.cons = c

        def pf(self):
            return ("", 0.0)

        def mfe(self):
            st, e, _ = canned_fold(self.frag, self.cons)
            return st, e

        def centroid(self):
            return ("", 0.0)

        def mean_bp_distance(self):
            return canned_fold(self.frag, self.cons)[2]

    rna = types.SimpleNamespace(fold_compound=FC, pf_fold=lambda s: ("", 0.0), PS_rna_plot_a=lambda *a: None)

    G = {"generated_by": "tests/golden/make_golden_motifs.py", "cases": []}
    for seq, line in CASES:
        class Nuc:
            def __init__(self, c):
                self.coordinate = c
        class Rec:
            pass
        rec = Rec()
        rec.seq = seq
        ns = dict(vars(sff))
        ns.update(filter2constraints=line, seq=seq, cur_record=rec, nuc_dict={k + 1: Nuc(k + 1) for k in range(len(seq))},
                  bond_order=[], bond_count=0)  # initialised a few lines above the block (ScanFold.py:1566-1567)
        buf = io.StringIO()
        case = {"sequence": seq, "structure_line": line}
        try:
            with contextlib.redirect_stdout(buf):
                exec(extract_src, ns)
        except Exception as e:  # the reference crashes on this input: record how
            case["error"] = type(e).__name__
            case["stdout"] = buf.getvalue()
            G["cases"].append(case)
            continue
        ex = ns["extracted_structure_list"]
        case["stdout"] = buf.getvalue()
        case["motifs"] = [{"count": e.structure_count, "sequence": e.sequence, "structure": e.structure, "i": e.i, "j": e.j}
                          for e in ex]
        usable = all(m["sequence"] for m in case["motifs"])
        if ex and usable:
            with tempfile.TemporaryDirectory() as d:
                cwd = os.getcwd()
                os.chdir(d)
                try:
                    ns2 = dict(vars(sff))
                    ns2.update(RNA=rna, extracted_structure_list=ex, structure_extract_file="x.gff3", name="rec|1",
                               temperature=37, algo="rnafold", type="mono", length=len(seq),
                               scramble=lambda frag, r, t: [frag[k % len(frag):] + frag[:k % len(frag)] for k in range(1, r + 1)],
                               energies=lambda sl, t, al: canned_energies(sl))
                    with contextlib.redirect_stdout(io.StringIO()):
                        exec(refold_src, ns2)
                    case["files"] = {fn: open(fn).read() for fn in sorted(os.listdir(d))}
                finally:
                    os.chdir(cwd)
        G["cases"].append(case)
    json.dump(G, open(OUT, "w"), indent=0)
    print("wrote", OUT, os.path.getsize(OUT), [c.get("error", len(c.get("motifs", []))) for c in G["cases"]])


if __name__ == "__main__":
    main()

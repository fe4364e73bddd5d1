"""Test helper: write a parameter set as a ViennaRNA "RNAfold parameter file v2.0" in the PUBLISHED file's layout —
free-energy section, then its `_enthalpies` twin, block comments before and after rows, dG / dH column pairs in
NINIO / ML_params / Misc and in the special-loop lists.  (The real rna_turner2004.par is absent from this machine;
this reproduces its format, not its numbers.)"""
import numpy as np

PAIRS = ["CG", "GC", "GU", "UG", "AU", "UA", "NS"]
INF = 10000000


def _tok(v, def_mask=False):
    if def_mask:
        return "DEF"
    return "INF" if v >= INF else ("-INF" if v <= -INF else str(int(v)))


def _rows(out, flat, per, mask=None, comments=None):
    flat = list(np.asarray(flat).reshape(-1))
    m = [False] * len(flat) if mask is None else list(np.asarray(mask).reshape(-1))
    for k in range(0, len(flat), per):
        line = " ".join("%6s" % _tok(v, d) for v, d in zip(flat[k:k + per], m[k:k + per]))
        if comments:
            line += "    /* %s */" % comments[(k // per) % len(comments)]
        out.append(line)


def par_text(rec, dH=None, def_fields=None):
    """rec / dH: numpy records of params.BLOB_DTYPE.  def_fields: {field: boolean mask} -> those entries become DEF."""
    def_fields = def_fields or {}
    out = ["## RNAfold parameter file v2.0", "",
           "/* This file contains energy parameters for RNA folding.            */",
           "/* written by tests/par_util.py in the layout of rna_turner2004.par */",
           "/* a comment that",
           "   spans two lines */", ""]
    mm = [("mismatch_hairpin", "mismatchH"), ("mismatch_interior", "mismatchI"),
          ("mismatch_interior_1n", "mismatch1nI"), ("mismatch_interior_23", "mismatch23I"),
          ("mismatch_multi", "mismatchM"), ("mismatch_exterior", "mismatchExt")]

    def both(name, emit):
        for suffix, r in (("", rec), ("_enthalpies", dH)):
            if r is None:
                continue
            out.append("# " + name + suffix)
            emit(r, def_fields if suffix == "" else {})
            out.append("")

    both("stack", lambda r, d: (out.append("/*  CG     GC     GU     UG     AU     UA     NS  */"),
                                 _rows(out, r["stack"][1:8, 1:8], 7, d.get("stack"), PAIRS)))
    for sec, field in mm:
        both(sec, lambda r, d, f=field: _rows(out, r[f][1:8], 5, d.get(f),
                                               ["%s,%s" % (p, b) for p in PAIRS for b in "NACGU"]))
    both("dangle5", lambda r, d: (out.append("/*  N      A      C      G      U  */"),
                                   _rows(out, r["dangle5"][1:8], 5, d.get("dangle5"), PAIRS)))
    both("dangle3", lambda r, d: (out.append("/*  N      A      C      G      U  */"),
                                   _rows(out, r["dangle3"][1:8], 5, d.get("dangle3"), PAIRS)))

    def int11(r, d):
        for a in range(1, 8):
            for b in range(1, 8):
                out.append("/* %s..%s */" % (PAIRS[a - 1], PAIRS[b - 1]))
                _rows(out, r["int11"][a, b], 5, None if "int11" not in d else d["int11"][a - 1, b - 1])
    both("int11", int11)

    def int21(r, d):
        for a in range(1, 8):
            for b in range(1, 8):
                for x in range(5):
                    out.append("/* %s.%s..%s */" % (PAIRS[a - 1], "NACGU"[x], PAIRS[b - 1]))
                    _rows(out, r["int21"][a, b, x], 5)
    both("int21", int21)

    def int22(r, d):
        for a in range(1, 7):
            for b in range(1, 7):
                for w in range(1, 5):
                    for x in range(1, 5):
                        out.append("/* %s.%s%s..%s */" % (PAIRS[a - 1], "NACGU"[w], "NACGU"[x], PAIRS[b - 1]))
                        _rows(out, r["int22"][a, b, w, x, 1:5, 1:5], 4)
    both("int22", int22)
    both("hairpin", lambda r, d: _rows(out, r["hairpin"], 10, d.get("hairpin")))
    both("bulge", lambda r, d: _rows(out, r["bulge"], 10))
    both("interior", lambda r, d: _rows(out, r["internal_loop"], 10))

    h = dH if dH is not None else np.zeros((), dtype=rec.dtype)
    out += ["# NINIO", "/* Ninio = MIN(max, m*|n1-n2| */", "/*       m   m_dH     max  */",
            "  %6d %6d %6d" % (rec["ninio"], h["ninio"], rec["max_ninio"]), ""]
    out += ["# ML_params", "/* F = cu*n_unpaired + cc + ci*loop_degree (+TermAU) */",
            "/*\t    cu\t    cu_dH\t    cc\t    cc_dH\t    ci\t    ci_dH  */",
            "\t%6d\t%6d\t%6d\t%6d\t%6d\t%6d" % (rec["MLbase"], h["MLbase"], rec["MLclosing"], h["MLclosing"],
                                               rec["MLintern"][1], h["MLintern"][1]), ""]
    out += ["# Misc", "/* all parameters are pairs of 'energy enthalpy' */",
            "/*    DuplexInit     TerminalAU   LXC  */",
            "   %d     %d     %d     %d     %.6f\t0.000000" % (rec["DuplexInit"], h["DuplexInit"], rec["TerminalAU"],
                                                              h["TerminalAU"], float(rec["lxc"])), ""]
    for sec, fseq, fe, fn in (("Triloops", "tri_seq", "tri_E", "n_tri"), ("Tetraloops", "tetra_seq", "tetra_E", "n_tetra"),
                              ("Hexaloops", "hexa_seq", "hexa_E", "n_hexa")):
        out.append("# " + sec)
        for k in range(int(rec[fn])):
            out.append("%s %6d %6d" % (bytes(rec[fseq][k]).rstrip(b"\0").decode(), rec[fe][k], h[fe][k]))
        out.append("")
    out += ["#END", ""]
    return "\n".join(out)


def synthetic_enthalpies(rec, seed=0):
    """An enthalpy record shaped like Turner's: dH ~ 3 x dG - noise (keeps INF where dG is INF)."""
    rng = np.random.default_rng(seed)
    dH = np.zeros((), dtype=rec.dtype)
    for f in ("stack", "hairpin", "bulge", "internal_loop", "mismatchI", "mismatchH", "mismatchM", "mismatch1nI",
              "mismatch23I", "mismatchExt", "dangle5", "dangle3", "int11", "int21", "int22", "tetra_E", "tri_E", "hexa_E"):
        g = rec[f].astype(np.int64)
        v = 3 * g - rng.integers(0, 40, size=g.shape) * 10
        dH[f] = np.where(np.abs(g) >= INF, g, v)
    dH["ninio"] = 320
    dH["MLbase"] = 0
    dH["MLclosing"] = 3000
    dH["MLintern"][:] = -220
    dH["TerminalAU"] = 370
    dH["DuplexInit"] = 360
    return dH

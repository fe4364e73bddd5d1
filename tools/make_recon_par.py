#!/usr/bin/env python3
"""Write scanfold_amd/params/rna_turner2004_recon.par (ViennaRNA ".par" v2.0 text format).

PROVENANCE — read this before trusting a number.
The reference delegates all thermodynamics to ViennaRNA (ScanFold-Scan.py:245,382-389), whose
compiled-in parameter file `rna_turner2004.par` is NOT present in the reference repository, in
this container, or on the GPU box, and there is no network.  This script
therefore *reconstructs* a Turner-2004-shaped parameter set:

  * small tables (stack, loop initiation, dangles, terminal/hairpin/interior mismatches,
    multiloop/ninio/misc constants, special hairpins) are written from the builder's memory of
    the published file; the N (unknown base) row/column and the NS (non-standard pair) block are
    derived with the "least stabilising of the known entries" rule the published file follows;
  * the large tables int11 / int21 / int22 (≈16 k integers) are produced by the documented
    nearest-neighbour RULES below, not by the published per-sequence measurements.

The set is internally consistent (symmetries of SURVEY.md A.5 hold, checked in tests) and is the
default blob of both the HIP library and the oracle, so GPU↔oracle parity is exact.  It is NOT
claimed to equal ViennaRNA's table entry for entry: **parity vs ViennaRNA is unpinned**.  A user
who owns a ViennaRNA installation passes its `rna_turner2004.par` through
`scanfold_amd.params.load_par(path)` (same text format) and gets the published values.
Free energies at 37 °C only; enthalpy sections are not written (temperature != 37 is a
"next" row).
"""
import os
import sys

INF = 10000000
BASES = "NACGU"
PAIRS = ["CG", "GC", "GU", "UG", "AU", "UA", "NS"]


def is_au(t):  # t: 1..6 ; AU/GU-type closing pair
    return 1 if t > 2 else 0


# ---- small tables, 4x4 cores [type][5' mismatch A,C,G,U][3' mismatch A,C,G,U] ----
STACK = [
    [-240, -330, -210, -140, -210, -210],
    [-330, -340, -250, -150, -220, -240],
    [-210, -250, 130, -50, -140, -130],
    [-140, -150, -50, 30, -60, -100],
    [-210, -220, -140, -60, -110, -90],
    [-210, -240, -130, -100, -90, -130],
]

MM_HAIRPIN = {
    "CG": [[-150, -150, -140, -150], [-100, -110, -100, -80], [-230, -150, -240, -150], [-100, -140, -100, -210]],
    "GC": [[-110, -150, -130, -150], [-110, -70, -110, -50], [-250, -150, -220, -150], [-110, -100, -110, -160]],
    "GU": [[20, -50, -30, -50], [-10, -20, -10, -20], [-100, -50, -110, -50], [-10, -30, -10, -100]],
    "UG": [[-50, -30, -60, -30], [-20, -10, -20, 0], [-90, -30, -110, -30], [-20, -10, -20, -90]],
    "AU": [[-30, -50, -30, -50], [-10, -20, -10, -20], [-120, -30, -110, -30], [-10, -30, -10, -100]],
    "UA": [[-50, -30, -50, -30], [-20, -10, -20, 0], [-150, -30, -150, -30], [-20, -10, -20, -80]],
}

MM_MULTI = {
    "CG": [[-110, -110, -160, -110], [-150, -70, -150, -100], [-130, -110, -140, -110], [-150, -50, -150, -70]],
    "GC": [[-150, -100, -140, -100], [-150, -110, -150, -140], [-140, -100, -160, -100], [-150, -80, -150, -120]],
    "GU": [[-100, -70, -50, -70], [-80, -60, -80, -60], [-110, -70, -80, -70], [-80, -50, -80, -50]],
    "UG": [[-30, -60, -60, -60], [-100, -70, -100, -80], [-80, -60, -80, -60], [-80, -60, -80, -60]],
    "AU": [[-100, -70, -110, -70], [-80, -60, -80, -60], [-110, -70, -120, -70], [-80, -50, -80, -50]],
    "UA": [[-80, -100, -80, -100], [-60, -70, -60, -70], [-80, -100, -80, -100], [-60, -80, -60, -80]],
}


def mm_interior_core(t, ga, ag, gg, uu):
    base = 70 * is_au(t)
    m = [[base] * 4 for _ in range(4)]
    m[0][2] += ag  # 5' A, 3' G
    m[2][0] += ga  # 5' G, 3' A
    m[2][2] += gg
    m[3][3] += uu
    return m


DANGLE5 = {"CG": [-50, -30, -20, -10], "GC": [-20, -30, 0, 0], "GU": [-30, -30, -40, -20],
           "UG": [-30, -10, -20, -20], "AU": [-30, -30, -40, -20], "UA": [-30, -10, -20, -20]}
DANGLE3 = {"CG": [-110, -40, -130, -60], "GC": [-170, -80, -170, -120], "GU": [-70, -10, -70, -10],
           "UG": [-80, -50, -80, -60], "AU": [-70, -10, -70, -10], "UA": [-80, -50, -80, -60]}

HAIRPIN = [INF, INF, INF, 540, 560, 570, 540, 600, 550, 640, 650, 660, 670, 680, 690, 690, 700, 710, 710,
           720, 720, 730, 730, 740, 740, 750, 750, 750, 760, 760, 770]
BULGE = [INF, 380, 280, 320, 360, 400, 440, 459, 470, 480, 490, 500, 510, 520, 530, 540, 540, 550, 550,
         560, 570, 570, 580, 580, 580, 590, 590, 600, 600, 600, 610]
INTERIOR = [INF, INF, 100, 100, 110, 200, 200, 210, 230, 240, 250, 260, 270, 280, 290, 290, 300, 310,
            310, 320, 330, 330, 340, 340, 350, 350, 350, 360, 360, 370, 370]

# (loop string incl. closing pair, dG37, dH) — the dH column is carried for format fidelity only
TETRA = [("CAACGG", 550, 690), ("CCAAGG", 330, -1030), ("CCACGG", 370, -330), ("CCCAGG", 340, -890),
         ("CCGAGG", 350, -660), ("CCGCGG", 360, -750), ("CCUAGG", 370, -350), ("CCUCGG", 250, -1390),
         ("CUAAGG", 360, -760), ("CUACGG", 280, -1070), ("CUCAGG", 370, -660), ("CUCCGG", 270, -1290),
         ("CUGCGG", 280, -1070), ("CUUAGG", 350, -620), ("CUUCGG", 370, -1530), ("CUUUGG", 370, -680)]
TRI = [("CAACG", 680, 2370), ("GUUAC", 690, 1080)]
HEXA = [("ACAGUACU", 280, -1680), ("ACAGUGAU", 360, -1140), ("ACAGUGCU", 290, -1280),
        ("ACAGUGUU", 180, -1540)]


def expand5(core):
    """4x4 core -> 5x5 with N row/col = max (least stabilising) of the known entries."""
    m = [[0] * 5 for _ in range(5)]
    for a in range(4):
        for b in range(4):
            m[a + 1][b + 1] = core[a][b]
    for b in range(1, 5):
        m[0][b] = max(core[a][b - 1] for a in range(4))
    for a in range(1, 5):
        m[a][0] = max(core[a - 1])
    m[0][0] = max(max(r) for r in core)
    return m


def with_ns(blocks):
    """6 blocks (any nesting of lists) -> 7 blocks with NS = elementwise max."""
    def emax(xs):
        if isinstance(xs[0], list):
            return [emax([x[i] for x in xs]) for i in range(len(xs[0]))]
        return max(xs)
    return blocks + [emax(blocks)]


def mm_mismatch_bonus(m5, m3):
    """first-mismatch bonus used by the int22 rule; m5/m3 in 1..4 (A,C,G,U)"""
    return {(3, 1): -100, (1, 3): -80, (4, 4): -60, (3, 3): -50}.get((m5, m3), 0)


def int11_rule(t1, t2, x, y):
    e = 50 + 70 * (is_au(t1) + is_au(t2))
    if x == 3 and y == 3:
        e -= 190 if (is_au(t1) + is_au(t2)) == 0 else 140
    elif x == 1 and y == 1 and is_au(t1) + is_au(t2) == 0:
        e += 40
    elif x == 4 and y == 4:
        e -= 10
    return e


def int21_rule(t1, t2, x, z, y):
    # x = lone nt on the 1-side, (z, y) = the two nts on the 2-side, y next to pair t1
    e = 230 + 70 * (is_au(t1) + is_au(t2))
    if x == 3 and y == 3:
        e -= 120
    if x == 3 and z == 3:
        e -= 120
    return e


def int22_rule(t1, t2, w, x, y, z):
    # int22[t1][t2][si1=w][sp1=x][sq1=y][sj1=z]; mismatch next to t1 is (w,z), next to t2 is (y,x)
    e = 120 + 70 * (is_au(t1) + is_au(t2))
    e += mm_mismatch_bonus(w, z) + mm_mismatch_bonus(y, x)
    return e


def fmt(v):
    return "INF" if v >= INF else str(v)


def rows(out, flat, per=5, comment=None):
    for k in range(0, len(flat), per):
        line = " ".join("%6s" % fmt(v) for v in flat[k:k + per])
        out.append(line + ("    /* %s */" % comment[k // per] if comment else ""))


def main(path):
    out = ["## RNAfold parameter file v2.0", "",
           "/* RECONSTRUCTED Turner-2004-shaped set written by tools/make_recon_par.py.      */",
           "/* NOT the published rna_turner2004.par: int11/int21/int22 are rule-generated.   */",
           "/* Free energies at 37 C only (no enthalpy sections). See that script's header. */", ""]
    # stack (7x7 incl. NS)
    st = [r[:] for r in STACK]
    ns_col = [max(r) for r in st]
    st7 = [st[i] + [ns_col[i]] for i in range(6)]
    st7.append([max(st7[i][j] for i in range(6)) for j in range(7)])
    out.append("# stack")
    out.append("/*  CG     GC     GU     UG     AU     UA     NS  */")
    rows(out, [v for r in st7 for v in r], per=7, comment=PAIRS)
    out.append("")

    def mm_section(name, cores):
        blocks = with_ns([expand5(cores[p]) for p in PAIRS[:6]])
        out.append("# " + name)
        for pi, blk in enumerate(blocks):
            rows(out, [v for r in blk for v in r], per=5,
                 comment=["%s,%s" % (PAIRS[pi], b) for b in BASES])
        out.append("")

    mm_section("mismatch_hairpin", MM_HAIRPIN)
    mm_section("mismatch_interior",
               {p: mm_interior_core(i + 1, -100, -80, -100, -60) for i, p in enumerate(PAIRS[:6])})
    mm_section("mismatch_interior_1n",
               {p: mm_interior_core(i + 1, 0, 0, 0, 0) for i, p in enumerate(PAIRS[:6])})
    mm_section("mismatch_interior_23",
               {p: mm_interior_core(i + 1, -110, -50, -70, -30) for i, p in enumerate(PAIRS[:6])})
    mm_section("mismatch_multi", MM_MULTI)
    mm_section("mismatch_exterior", MM_MULTI)

    for name, tab in (("dangle5", DANGLE5), ("dangle3", DANGLE3)):
        blocks = []
        for p in PAIRS[:6]:
            blocks.append([max(tab[p])] + tab[p])
        blocks = with_ns(blocks)
        out.append("# " + name)
        out.append("/*  N      A      C      G      U  */")
        rows(out, [v for b in blocks for v in b], per=5, comment=PAIRS)
        out.append("")

    # int11: 7x7 pair combos x 5x5
    def full5(fn4, nd):
        """fn4(idx tuple in 1..4) -> nested 5^nd list with N entries = max over the known bases"""
        import itertools
        vals = {}
        for idx in itertools.product(range(5), repeat=nd):
            choices = [range(1, 5) if k == 0 else [k] for k in idx]
            vals[idx] = max(fn4(*c) for c in itertools.product(*choices))
        return [vals[idx] for idx in itertools.product(range(5), repeat=nd)]

    out.append("# int11")
    for a in range(1, 8):
        for b in range(1, 8):
            ta = range(1, 7) if a == 7 else [a]
            tb = range(1, 7) if b == 7 else [b]
            flat = full5(lambda x, y: max(int11_rule(p, q, x, y) for p in ta for q in tb), 2)
            out.append("/* %s..%s */" % (PAIRS[a - 1], PAIRS[b - 1]))
            rows(out, flat, per=5)
    out.append("")
    out.append("# int21")
    for a in range(1, 8):
        for b in range(1, 8):
            ta = range(1, 7) if a == 7 else [a]
            tb = range(1, 7) if b == 7 else [b]
            flat = full5(lambda x, z, y: max(int21_rule(p, q, x, z, y) for p in ta for q in tb), 3)
            out.append("/* %s..%s */" % (PAIRS[a - 1], PAIRS[b - 1]))
            rows(out, flat, per=5)
    out.append("")
    out.append("# int22")
    for a in range(1, 7):
        for b in range(1, 7):
            out.append("/* %s..%s */" % (PAIRS[a - 1], PAIRS[b - 1]))
            flat = [int22_rule(a, b, w, x, y, z) for w in range(1, 5) for x in range(1, 5)
                    for y in range(1, 5) for z in range(1, 5)]
            rows(out, flat, per=4)
    out.append("")
    for name, arr in (("hairpin", HAIRPIN), ("bulge", BULGE), ("interior", INTERIOR)):
        out.append("# " + name)
        rows(out, arr, per=10)
        out.append("")
    # the three sections below keep the published file's column structure (dG and dH interleaved)
    out += ["# NINIO", "/* Ninio = MIN(max, m*|n1-n2| */", "/*       m   m_dH     max  */",
            "      60    320     300", ""]
    out += ["# ML_params", "/* F = cu*n_unpaired + cc + ci*loop_degree (+TermAU) */",
            "/*      cu   cu_dH      cc   cc_dH      ci   ci_dH  */",
            "       0       0     930    3000     -90    -220", ""]
    out += ["# Misc", "/* all parameters are pairs of 'energy enthalpy' */",
            "/*    DuplexInit     TerminalAU   LXC  */",
            "     410    360     50    370    107.856000   0", ""]
    out.append("# Triloops")
    out += ["%s %6d %6d" % t for t in TRI] + [""]
    out.append("# Tetraloops")
    out += ["%s %6d %6d" % t for t in TETRA] + [""]
    out.append("# Hexaloops")
    out += ["%s %6d %6d" % t for t in HEXA] + [""]
    out += ["# END", ""]
    with open(path, "w") as f:
        f.write("\n".join(out))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        here, "..", "scanfold_amd", "params", "rna_turner2004_recon.par")
    main(dst)
    print("wrote", os.path.normpath(dst))

"""A second, independent statement of the nearest-neighbour model — pure Python, in ENERGY space, for sequences short enough
to enumerate (test infrastructure only).

Why it exists (round-3 advice): at T != 37 the product (`ex()` in the HIP library), the oracle (`exact_energy()` in
oracle/sf_oracle.c) and the CPU twin derive the un-truncated Boltzmann weights from the same few lines of C, so a mistake they
share — the ninio cap, a field missing from the enthalpy record, the smoothing of the dangle terms — passes every
product-vs-oracle test.  This module shares no code with them: it reads the 37 C and enthalpy records of a ParamSet by field
name, rescales every entry itself,

    dG(T) = dH - (dH - dG37) * (T + 273.15) / 310.15                      (ViennaRNA get_scaled_params / get_boltzmann_factors [EXT])

adds up a structure's loops as real-valued energies (the C code multiplies per-loop weights instead), and sums
exp(-G / kT) over every structure of the sequence.  The dangle-like terms (dangle5 / dangle3 / mismatchM / mismatchExt) enter the
ensemble through ViennaRNA's SMOOTH() [EXT], restated below from the macro.  The MFE side uses the integer tables truncated
towards zero and clamps the dangle-like terms to <= 0, as ViennaRNA's get_scaled_params does.

Model: dangles = 2 (every stem of a multiloop / the exterior loop gets the mismatch of both neighbours,
a dangle at a sequence end), no lonely-pair filter, hairpins >= 3, interior loops <= 30 unpaired, special hairpins by string."""
import math

import numpy as np

K0 = 273.15
GASCONST = 1.98717  # cal / (K mol)
INF = 10000000
CODE = {"A": 1, "C": 2, "G": 3, "U": 4}
PAIR = {("C", "G"): 1, ("G", "C"): 2, ("G", "U"): 3, ("U", "G"): 4, ("A", "U"): 5, ("U", "A"): 6}
RTYPE = (0, 2, 1, 4, 3, 6, 5, 7)
_RESCALED = ("stack", "hairpin", "bulge", "internal_loop", "mismatchI", "mismatchH", "mismatchM", "mismatch1nI", "mismatch23I",
             "mismatchExt", "dangle5", "dangle3", "int11", "int21", "int22", "ninio", "MLbase", "MLclosing", "MLintern",
             "TerminalAU", "tetra_E", "tri_E", "hexa_E")
_DANGLE_LIKE = ("dangle5", "dangle3", "mismatchM", "mismatchExt")


def smooth(x):
    """ViennaRNA's SMOOTH(X) [EXT]: 0 below -1.2283697 * 10, X above 0.8660254 * 10, a sin^2 blend between."""
    s = x / 10.0
    if s < -1.2283697:
        return 0.0
    if s > 0.8660254:
        return x
    return 10.0 * 0.38490018 * (math.sin(s - 0.34242663) + 1.0) ** 2


class Model:
    """mode "pf": real-valued rescaled energies, dangle-like terms smoothed;  mode "mfe": truncated integers, dangle-like <= 0."""

    def __init__(self, paramset, temperature_c, mode):
        assert mode in ("pf", "mfe")
        self.T = float(temperature_c)
        self.mode = mode
        self.kT = (self.T + K0) * GASCONST
        g37 = paramset.rec37
        dH = paramset.dH
        tempf = (self.T + K0) / (37.0 + K0)
        self.t = {}
        for f in _RESCALED:
            a37 = np.array(g37[f], dtype=float)  # (scalar fields become 0-d arrays: reshape(-1) below must be a view)
            if abs(self.T - 37.0) < 1e-12 or dH is None:
                v = a37.copy()
            else:
                h = np.array(dH[f], dtype=float)
                v = h - (h - a37) * tempf
            v = np.atleast_1d(v)
            flat, src = v.reshape(-1), np.atleast_1d(a37).reshape(-1)
            for k in range(flat.size):
                if abs(src[k]) >= INF:
                    flat[k] = src[k]
                elif mode == "mfe":
                    flat[k] = math.trunc(flat[k])
                    if f in _DANGLE_LIKE:
                        flat[k] = min(0.0, flat[k])
                elif f in _DANGLE_LIKE:
                    flat[k] = -smooth(-flat[k])
            self.t[f] = v if a37.ndim else v[0]
        self.max_ninio = float(g37["max_ninio"])
        self.seen = set()  # loop classes evaluated so far (the tests check that their sequences reach all of them)
        self.specials = {}
        for name, width in (("tetra", 6), ("tri", 5), ("hexa", 8)):
            n = int(g37["n_" + name])
            for k in range(n):
                s = bytes(g37[name + "_seq"][k]).split(b"\0")[0].decode()
                self.specials.setdefault(s[:width], float(self.t[name + "_E"][k]))  # first entry wins, as the look-up does

    # ---- loops ----
    def _tau(self, typ):
        return float(self.t["TerminalAU"]) if typ > 2 else 0.0

    def hairpin(self, seq, i, j):
        """pair (i, j), 1-based; loop string includes the closing pair"""
        size = j - i - 1
        typ = PAIR[(seq[i - 1], seq[j - 1])]
        assert 3 <= size <= 30
        e = float(self.t["hairpin"][size])
        loop = seq[i - 1:j]
        if size in (3, 4, 6) and loop in self.specials:
            self.seen.add("special hairpin %d" % size)
            return self.specials[loop]
        self.seen.add("hairpin 3" if size == 3 else "hairpin")
        if size == 3:
            return e + self._tau(typ)
        return e + float(self.t["mismatchH"][typ][CODE[seq[i]]][CODE[seq[j - 2]]])

    def interior(self, seq, i, j, p, q):
        typ = PAIR[(seq[i - 1], seq[j - 1])]
        typ2 = RTYPE[PAIR[(seq[p - 1], seq[q - 1])]]
        n1, n2 = p - i - 1, j - q - 1
        si1, sj1, sp1, sq1 = CODE[seq[i]], CODE[seq[j - 2]], CODE[seq[p - 2]], CODE[seq[q]]
        nl, ns = max(n1, n2), min(n1, n2)
        t = self.t
        self.seen.add("stack" if nl == 0 else "bulge %d" % min(nl, 2) if ns == 0 else "%dx%d" % (min(ns, 3), min(nl, 4)))
        if nl == 0:
            return float(t["stack"][typ][typ2])
        if ns == 0:
            e = float(t["bulge"][nl])
            return e + float(t["stack"][typ][typ2]) if nl == 1 else e + self._tau(typ) + self._tau(typ2)
        asym = min(self.max_ninio, (nl - ns) * float(t["ninio"]))
        if ns == 1:
            if nl == 1:
                return float(t["int11"][typ][typ2][si1][sj1])
            if nl == 2:
                return float(t["int21"][typ][typ2][si1][sq1][sj1]) if n1 == 1 else float(t["int21"][typ2][typ][sq1][si1][sp1])
            return float(t["internal_loop"][nl + 1]) + asym + float(t["mismatch1nI"][typ][si1][sj1]) + \
                float(t["mismatch1nI"][typ2][sq1][sp1])
        if ns == 2 and nl == 2:
            return float(t["int22"][typ][typ2][si1][sp1][sq1][sj1])
        if ns == 2 and nl == 3:
            return float(t["internal_loop"][5]) + float(t["ninio"]) + float(t["mismatch23I"][typ][si1][sj1]) + \
                float(t["mismatch23I"][typ2][sq1][sp1])
        return float(t["internal_loop"][nl + ns]) + asym + float(t["mismatchI"][typ][si1][sj1]) + float(t["mismatchI"][typ2][sq1][sp1])

    def _stem(self, table, typ, five, three):
        """a stem's end term in a multiloop ("mismatchM") or the exterior loop ("mismatchExt"); five / three = neighbour codes or None"""
        if five is not None and three is not None:
            e = float(self.t[table][typ][five][three])
        elif five is not None:
            e = float(self.t["dangle5"][typ][five])
        elif three is not None:
            e = float(self.t["dangle3"][typ][three])
        else:
            e = 0.0
        return e + self._tau(typ)

    def energy(self, seq, pt, sc=None):
        """pt: 1-based partner table (0 = unpaired), pt[0] unused.  dcal/mol.  sc: Deigan pseudo-energies per nucleotide (dcal,
        0-based): a stack (i, j) on (i+1, j-1) pays those of its four nucleotides (ScanFold.py:522-544)."""
        n = len(seq)
        total = 0.0
        i = 1
        while i <= n:  # exterior loop
            if pt[i] == 0:
                i += 1
                continue
            j = pt[i]
            typ = PAIR[(seq[i - 1], seq[j - 1])]
            total += self._stem("mismatchExt", typ, CODE[seq[i - 2]] if i > 1 else None, CODE[seq[j]] if j < n else None)
            i = j + 1
        for i in range(1, n + 1):
            j = pt[i]
            if j <= i:
                continue
            inner, k, unpaired = [], i + 1, 0
            while k < j:
                if pt[k] == 0:
                    unpaired += 1
                    k += 1
                else:
                    inner.append((k, pt[k]))
                    k = pt[k] + 1
            if not inner:
                total += self.hairpin(seq, i, j)
            elif len(inner) == 1:
                total += self.interior(seq, i, j, inner[0][0], inner[0][1])
                if sc is not None and inner[0] == (i + 1, j - 1):
                    total += sc[i - 1] + sc[i] + sc[j - 2] + sc[j - 1]
            else:
                typ = PAIR[(seq[i - 1], seq[j - 1])]
                self.seen.add("multiloop")
                total += float(self.t["MLclosing"]) + unpaired * float(self.t["MLbase"])
                # the closing pair seen from inside the loop: reversed type, neighbours j-1 (5' side) and i+1 (3' side)
                total += self._stem("mismatchM", RTYPE[typ], CODE[seq[j - 2]], CODE[seq[i]]) + float(self.t["MLintern"][RTYPE[typ]])
                for (p, q) in inner:
                    t2 = PAIR[(seq[p - 1], seq[q - 1])]
                    total += self._stem("mismatchM", t2, CODE[seq[p - 2]], CODE[seq[q]]) + float(self.t["MLintern"][t2])
        return total


def structures(seq, min_hairpin=3, max_loop=30):
    """Every secondary structure of seq as a 1-based partner table (canonical + GU pairs, interior loops <= max_loop unpaired)."""
    n = len(seq)

    def ok_loops(pt):
        for i in range(1, n + 1):
            j = pt[i]
            if j > i:
                inner, k, unp = 0, i + 1, 0
                while k < j:
                    if pt[k] == 0:
                        unp += 1
                        k += 1
                    else:
                        inner += 1
                        k = pt[k] + 1
                if inner == 1 and unp > max_loop:
                    return False
        return True

    def rec(i, j):
        """structures on [i, j] as lists of pairs"""
        if j - i < min_hairpin + 1:
            yield []
            return
        for rest in rec(i + 1, j):  # i unpaired
            yield rest
        for k in range(i + min_hairpin + 1, j + 1):  # i pairs with k
            if (seq[i - 1], seq[k - 1]) in PAIR:
                for inside in rec(i + 1, k - 1):
                    for after in rec(k + 1, j):
                        yield [(i, k)] + inside + after

    for pairs in rec(1, n):
        pt = [0] * (n + 2)
        for (a, b) in pairs:
            pt[a], pt[b] = b, a
        if ok_loops(pt):
            yield pt


def admissible(pt, cons):
    """Hard constraint in ViennaRNA's dot-bracket notation with its default, non-enforcing options: `x` stays
    unpaired; `<` / `>` may pair with a partner downstream / upstream only; a bracket pair may pair with each other only and
    no pair may cross it; `|` and `.` change nothing.  Stated on whole structures (the oracle states it per cell of its DP)."""
    n = len(cons)
    brackets, stack = [], []
    for k, ch in enumerate(cons, 1):
        if ch == "(":
            stack.append(k)
        elif ch == ")":
            brackets.append((stack.pop(), k))
    assert not stack
    for i in range(1, n + 1):
        j = pt[i]
        if j <= i:
            continue
        ci, cj = cons[i - 1], cons[j - 1]
        if ci == "x" or cj == "x" or ci == ">" or cj == "<":
            return False
        if (ci in "()" or cj in "()") and (i, j) not in brackets:
            return False
        for (a, b) in brackets:
            if (i < a < j < b) or (a < i < b < j):
                return False
    return True


def mfe_with_shape(paramset, seq, temperature_c, sc):
    """Minimum over every structure of the integer-table energy with the Deigan stack terms -> (dcal, dot-bracket)."""
    m = Model(paramset, temperature_c, "mfe")
    best, best_pt = None, None
    for pt in structures(seq):
        e = m.energy(seq, pt, sc)
        if best is None or e < best:
            best, best_pt = e, pt
    n = len(seq)
    return int(round(best)), "".join("." if best_pt[i] == 0 else ("(" if best_pt[i] > i else ")") for i in range(1, n + 1))


def ensemble(paramset, seq, temperature_c, cons=None):
    """(ensemble free energy in kcal/mol, mean base-pair distance, number of structures) at T from the un-truncated energies,
    (MFE in dcal/mol, its structure) from the truncated ones."""
    pf, mfe = Model(paramset, temperature_c, "pf"), Model(paramset, temperature_c, "mfe")
    n = len(seq)
    z, count = 0.0, 0
    pp = {}
    best, best_pt = None, None
    for pt in structures(seq):
        if cons is not None and not admissible(pt, cons):
            continue
        w = math.exp(-pf.energy(seq, pt) * 10.0 / pf.kT)
        z += w
        count += 1
        for i in range(1, n + 1):
            if pt[i] > i:
                pp[(i, pt[i])] = pp.get((i, pt[i]), 0.0) + w
        e = mfe.energy(seq, pt)
        if best is None or e < best:
            best, best_pt = e, pt
    dg = -pf.kT * math.log(z) / 1000.0
    dist = sum(2.0 * (w / z) * (1.0 - w / z) for w in pp.values())
    db = "".join("." if best_pt[i] == 0 else ("(" if best_pt[i] > i else ")") for i in range(1, n + 1))
    return dg, dist, count, int(round(best)), db, pf.seen

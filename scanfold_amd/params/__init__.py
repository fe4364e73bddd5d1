"""Energy-parameter sets: ViennaRNA ".par" v2.0 text -> the binary blob of include/sf_params_blob.h.

The reference never reads parameters itself; they live inside ViennaRNA, reached through
`RNA.md()` / `RNA.fold_compound(seq, md)` (ScanFold-Scan.py:70-71,382; ScanFoldFunctions.py:776-786).
Here they are plain data handed to `sf_params_load` (include/scanfold_hip.h).

`default_params()` loads the reconstructed set shipped in this directory (see
tools/make_recon_par.py for its provenance: parity vs ViennaRNA's own table is unpinned);
`load_par(path)` reads any file in the same text format, e.g. a real `rna_turner2004.par`.
"""
import itertools
import os
import re

import numpy as np

INF = 10000000
MAX_SPECIAL = 40
_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_PAR = os.path.join(_HERE, "rna_turner2004_recon.par")

# numpy mirror of struct sf_params_blob (include/sf_params_blob.h); align=True follows the C ABI
BLOB_DTYPE = np.dtype([
    ("magic", "<u4"), ("version", "<u4"),
    ("temperature", "<f8"), ("lxc", "<f8"),
    ("stack", "<i4", (8, 8)),
    ("hairpin", "<i4", (31,)), ("bulge", "<i4", (31,)), ("internal_loop", "<i4", (31,)),
    ("mismatchI", "<i4", (8, 5, 5)), ("mismatchH", "<i4", (8, 5, 5)), ("mismatchM", "<i4", (8, 5, 5)),
    ("mismatch1nI", "<i4", (8, 5, 5)), ("mismatch23I", "<i4", (8, 5, 5)), ("mismatchExt", "<i4", (8, 5, 5)),
    ("dangle5", "<i4", (8, 5)), ("dangle3", "<i4", (8, 5)),
    ("int11", "<i4", (8, 8, 5, 5)),
    ("int21", "<i4", (8, 8, 5, 5, 5)),
    ("int22", "<i4", (8, 8, 5, 5, 5, 5)),
    ("ninio", "<i4"), ("max_ninio", "<i4"),
    ("MLbase", "<i4"), ("MLclosing", "<i4"), ("MLintern", "<i4", (8,)),
    ("TerminalAU", "<i4"), ("DuplexInit", "<i4"),
    ("n_tetra", "<i4"), ("n_tri", "<i4"), ("n_hexa", "<i4"), ("pad0", "<i4"),
    ("tetra_seq", "S8", (MAX_SPECIAL,)), ("tetra_E", "<i4", (MAX_SPECIAL,)),
    ("tri_seq", "S8", (MAX_SPECIAL,)), ("tri_E", "<i4", (MAX_SPECIAL,)),
    ("hexa_seq", "S12", (MAX_SPECIAL,)), ("hexa_E", "<i4", (MAX_SPECIAL,)),
], align=True)

MAGIC = 0x31504653
VERSION = 1

_MM_SECTIONS = {
    "mismatch_hairpin": "mismatchH", "mismatch_interior": "mismatchI",
    "mismatch_interior_1n": "mismatch1nI", "mismatch_interior_23": "mismatch23I",
    "mismatch_multi": "mismatchM", "mismatch_exterior": "mismatchExt",
}


K0 = 273.15
T_MEASURE = 37.0 + K0

# free-energy field -> what is rescaled with it (all fields that have an enthalpy twin in a .par file)
_RESCALED = ("stack", "hairpin", "bulge", "internal_loop", "mismatchI", "mismatchH", "mismatchM", "mismatch1nI",
             "mismatch23I", "mismatchExt", "dangle5", "dangle3", "int11", "int21", "int22", "ninio", "MLbase",
             "MLclosing", "MLintern", "TerminalAU", "DuplexInit", "tetra_E", "tri_E", "hexa_E")


class ParamSet:
    """One parameter set; `.rec` is a numpy record with the fields of sf_params_blob (free energies at
    `.temperature`), `.dH` — when the source file had `*_enthalpies` sections — the same record holding enthalpies,
    `.rec37` the free energies at 37 C the set was read with."""

    def_substituted = {}  # section -> number of "DEF" entries the parser filled from the shipped reconstructed table

    def __init__(self, rec, source, dH=None, rec37=None, lxc37=None):
        self.rec = rec
        self.source = source
        self.dH = dH
        self.rec37 = rec37 if rec37 is not None else rec
        self.lxc37 = float(rec["lxc"]) if lxc37 is None else lxc37

    @staticmethod
    def _blob_of(rec):
        out = np.zeros((), dtype=BLOB_DTYPE)  # deterministic padding bytes (numpy leaves them undefined on copies)
        for f in BLOB_DTYPE.names:
            out[f] = rec[f]
        return out.tobytes()

    def blob(self):
        return self._blob_of(self.rec)

    def rescale_blobs(self):
        """(blob at 37 C, enthalpy blob) for sf_params_load_rescaled when this set was rescaled away from 37 C, else
        (None, None): the Boltzmann weights are then built from the exact rescaled doubles, not from the truncated
        integers of .blob() (ViennaRNA get_boltzmann_factors [EXT])."""
        if self.dH is None or self.rec37 is self.rec or abs(self.temperature - 37.0) < 1e-12:
            return None, None
        dh = self.dH.copy()
        dh["magic"], dh["version"] = self.rec["magic"], self.rec["version"]
        return self._blob_of(self.rec37), self._blob_of(dh)

    @property
    def temperature(self):
        return float(self.rec["temperature"])

    def copy(self):
        return ParamSet(self.rec.copy(), self.source, None if self.dH is None else self.dH.copy(),
                        None if self.rec37 is self.rec else self.rec37.copy(), self.lxc37)

    def at_temperature(self, temperature_c):
        """The set rescaled to another temperature the way ViennaRNA's get_scaled_params does it [EXT]:
        dG(T) = dH - (dH - dG37) * (T + K0) / (37 + K0), computed in double and truncated towards zero (the C code
        assigns the double to an int); lxc scales linearly; the MFE model later clamps dangles / multiloop and
        exterior mismatches to <= 0 (sf_params_load).  Needs the enthalpy sections of a real .par file: the
        reconstructed default set has none (RNA.md().temperature, ScanFold-Scan.py:70-71; ScanFoldFunctions.py:776-777)."""
        t = float(temperature_c)
        if abs(t - 37.0) < 1e-12:
            base = ParamSet(self.rec37.copy(), self.source, self.dH, None, self.lxc37)
            return base
        if self.dH is None:
            raise NotImplementedError(
                "folding temperature %s C: parameter set %s has no enthalpy tables (it is valid at 37 C only).  Pass ViennaRNA's "
                "own parameter file — `--params <ViennaRNA prefix>/share/ViennaRNA/rna_turner2004.par` on the command line, "
                "params.load_par(path) in code — whose *_enthalpies sections the rescale needs; the shipped default is a "
                "37 C reconstruction without them" % (temperature_c, self.source))
        tempf = (t + K0) / T_MEASURE
        out = self.rec37.copy()
        for f in _RESCALED:
            g37 = self.rec37[f].astype(np.float64)
            dh = self.dH[f].astype(np.float64)
            v = np.trunc(dh - (dh - g37) * tempf)
            v = np.where(np.abs(self.rec37[f]) >= INF, self.rec37[f], v)  # INF stays INF
            out[f] = v.astype(np.int64)
        out["lxc"] = self.lxc37 * tempf
        out["temperature"] = t
        return ParamSet(out, self.source, self.dH, self.rec37, self.lxc37)


_DEF = -(1 << 40)  # marker of a "DEF" token: keep the default value of that entry


def _tok_int(t):
    if t == "INF":
        return INF
    if t == "-INF":
        return -INF
    if t == "DEF":
        return _DEF
    return int(t)


def _fill_sections(sections, rec, suffix, source, defaults, substituted=None):
    """Fill `rec` from the sections `<name><suffix>` (suffix "" = free energies, "_enthalpies" = enthalpies).
    A "DEF" token keeps the entry of `defaults` (ViennaRNA keeps its compiled-in value there; here the shipped
    default set stands in, which is only right for the entries the two sets share)."""
    def ints(name, n):
        toks = " ".join(sections[name + suffix]).split()
        if len(toks) != n:
            raise ValueError("%s: section '%s' has %d values, expected %d" % (source, name + suffix, len(toks), n))
        return np.array([_tok_int(t) for t in toks], dtype=np.int64)

    def put(field, index, values):
        dst = rec[field][index]
        if (values == _DEF).any():
            if defaults is None:
                raise ValueError("%s: DEF entries in '%s' but no default set to take them from" % (source, field))
            if substituted is not None:
                substituted[field + suffix] = substituted.get(field + suffix, 0) + int((values == _DEF).sum())
            values = np.where(values == _DEF, defaults[field][index], values)
        rec[field][index] = values.reshape(dst.shape)

    put("stack", np.s_[1:8, 1:8], ints("stack", 49).reshape(7, 7))
    for sec, field in _MM_SECTIONS.items():
        put(field, np.s_[1:8], ints(sec, 175).reshape(7, 5, 5))
    put("dangle5", np.s_[1:8], ints("dangle5", 35).reshape(7, 5))
    put("dangle3", np.s_[1:8], ints("dangle3", 35).reshape(7, 5))
    put("int11", np.s_[1:8, 1:8], ints("int11", 49 * 25).reshape(7, 7, 5, 5))
    put("int21", np.s_[1:8, 1:8], ints("int21", 49 * 125).reshape(7, 7, 5, 5, 5))
    core = ints("int22", 36 * 256).reshape(6, 6, 4, 4, 4, 4)
    if (core == _DEF).any():
        if defaults is None:
            raise ValueError("%s: DEF entries in 'int22' but no default set to take them from" % source)
        if substituted is not None:
            substituted["int22" + suffix] = substituted.get("int22" + suffix, 0) + int((core == _DEF).sum())
        core = np.where(core == _DEF, defaults["int22"][1:7, 1:7, 1:5, 1:5, 1:5, 1:5], core)
    i22 = np.zeros((8, 8, 5, 5, 5, 5), dtype=np.int64)
    i22[1:7, 1:7, 1:5, 1:5, 1:5, 1:5] = core
    # entries with an unknown base (index 0) / non-standard pair (7): least stabilising known entry
    for ax in (2, 3, 4, 5):
        sl = [slice(None)] * 6
        sl[ax] = slice(1, 5)
        dst = [slice(None)] * 6
        dst[ax] = 0
        i22[tuple(dst)] = i22[tuple(sl)].max(axis=ax)
    i22[7, :] = i22[1:7, :].max(axis=0)
    i22[:, 7] = i22[:, 1:7].max(axis=1)
    i22[0, :] = 0
    i22[:, 0] = 0
    rec["int22"] = i22
    put("hairpin", np.s_[:], ints("hairpin", 31))
    put("bulge", np.s_[:], ints("bulge", 31))
    put("internal_loop", np.s_[:], ints("interior", 31))


def parse_par_text(text, source="<string>", defaults="shipped"):
    """ViennaRNA "RNAfold parameter file v2.0" text -> ParamSet.  Sections are `# name` lines; `/* ... */` comments
    (also spanning lines) and `##` lines are skipped; `_enthalpies` sections, when all present, are kept for
    ParamSet.at_temperature; "DEF" entries fall back to `defaults` ("shipped": the set of this package, None: error)."""
    if not text.lstrip().startswith("## RNAfold parameter file v2.0"):
        raise ValueError("%s: not a 'RNAfold parameter file v2.0'" % source)
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    sections = {}
    cur = None
    for line in text.splitlines():
        line = line.strip()
        if not line or line.startswith("##"):
            continue
        if line.startswith("#"):
            cur = line[1:].strip()
            if cur == "END":
                break
            sections[cur] = []
            continue
        if cur is not None:
            sections[cur].append(line)

    def_rec = None
    if defaults == "shipped":
        if os.path.abspath(str(source)) != os.path.abspath(DEFAULT_PAR) and "DEF" in text.split():
            def_rec = default_params().rec37
    elif defaults is not None:
        def_rec = defaults.rec37

    rec = np.zeros((), dtype=BLOB_DTYPE)
    rec["magic"] = MAGIC
    rec["version"] = VERSION
    rec["temperature"] = 37.0

    table_secs = ["stack", "hairpin", "bulge", "interior", "dangle5", "dangle3", "int11", "int21", "int22"] + \
        list(_MM_SECTIONS)
    need = table_secs + ["NINIO", "ML_params", "Misc"]
    missing = [s for s in need if s not in sections]
    if missing:
        raise ValueError("%s: missing sections %s" % (source, missing))
    substituted = {}
    _fill_sections(sections, rec, "", source, def_rec, substituted if defaults == "shipped" else None)

    have_dh = all((s + "_enthalpies") in sections for s in table_secs)
    dH = None
    if have_dh:
        dH = np.zeros((), dtype=BLOB_DTYPE)
        _fill_sections(sections, dH, "_enthalpies", source, None)

    def scalar(tok, field, index=None):
        v = _tok_int(tok)
        if v == _DEF:
            if def_rec is None:
                raise ValueError("%s: DEF entry for %s but no default set" % (source, field))
            v = int(def_rec[field] if index is None else def_rec[field][index])
            if defaults == "shipped":
                substituted[field] = substituted.get(field, 0) + 1
        return v

    nin = " ".join(sections["NINIO"]).split()       # m  m_dH  max
    rec["ninio"] = scalar(nin[0], "ninio")
    rec["max_ninio"] = scalar(nin[2], "max_ninio")
    ml = " ".join(sections["ML_params"]).split()    # cu cu_dH cc cc_dH ci ci_dH
    rec["MLbase"] = scalar(ml[0], "MLbase")
    rec["MLclosing"] = scalar(ml[2], "MLclosing")
    rec["MLintern"][:] = scalar(ml[4], "MLintern", 1)
    misc = " ".join(sections["Misc"]).split()       # DuplexInit dH TerminalAU dH [lxc lxc_dH]
    rec["DuplexInit"] = scalar(misc[0], "DuplexInit")
    rec["TerminalAU"] = scalar(misc[2], "TerminalAU")
    rec["lxc"] = float(misc[4]) if len(misc) > 4 else 107.856
    if dH is not None:
        dH["ninio"] = _tok_int(nin[1])
        dH["MLbase"] = _tok_int(ml[1])
        dH["MLclosing"] = _tok_int(ml[3])
        dH["MLintern"][:] = _tok_int(ml[5])
        dH["DuplexInit"] = _tok_int(misc[1])
        dH["TerminalAU"] = _tok_int(misc[3])

    for sec, fseq, fe, fn, ln in (("Tetraloops", "tetra_seq", "tetra_E", "n_tetra", 6),
                                  ("Triloops", "tri_seq", "tri_E", "n_tri", 5),
                                  ("Hexaloops", "hexa_seq", "hexa_E", "n_hexa", 8)):
        k = 0
        for line in sections.get(sec, []):
            parts = line.split()
            if len(parts) < 2:
                continue
            if len(parts[0]) != ln:
                raise ValueError("%s: %s entry '%s' must have %d characters" % (source, sec, parts[0], ln))
            if k >= MAX_SPECIAL:
                raise ValueError("%s: more than %d %s" % (source, MAX_SPECIAL, sec))
            rec[fseq][k] = parts[0].encode()
            rec[fe][k] = _tok_int(parts[1])
            if dH is not None:
                if len(parts) < 3:
                    dH = None  # a special loop without an enthalpy: the set cannot be rescaled
                else:
                    dH[fe][k] = _tok_int(parts[2])
            k += 1
        rec[fn] = k
    ps = ParamSet(rec, source, dH)
    # "DEF" entries of a user's file: ViennaRNA would keep its compiled-in (published) value there, this parser can only
    # put the shipped RECONSTRUCTED value — a mixed table.  Say so, and let --require-published-params refuse it.
    ps.def_substituted = substituted
    if substituted:
        import sys
        print("scanfold_amd: WARNING %s has %d DEF entries (sections: %s); they were filled from the RECONSTRUCTED default "
              "table %s, not from ViennaRNA's compiled-in values" % (source, sum(substituted.values()),
                                                                     ", ".join(sorted(substituted)), DEFAULT_PAR),
              file=sys.stderr)
    return ps


def load_par(path):
    with open(path, "r") as f:
        return parse_par_text(f.read(), source=path)


def is_reconstructed(paramset):
    """True for the set shipped with this package and anything derived from it (random_params keeps the source tag)."""
    return "recon" in os.path.basename(str(paramset.source))


_warned = False


def warn_if_reconstructed(paramset, stream=None):
    """One line on stderr per process when folds run on the reconstructed table: its int11 / int21 / int22 entries are
    rule-generated and its small tables were typed from memory (tools/make_recon_par.py), so energies, structures and
    z-scores differ from ViennaRNA's for windows with 1x1, 2x1 or 2x2 interior loops."""
    global _warned
    import sys
    if is_reconstructed(paramset) and not _warned:
        _warned = True
        print("scanfold_amd: WARNING energy parameters = %s, a RECONSTRUCTION of Turner 2004, not ViennaRNA's "
              "rna_turner2004.par: results are not comparable with ViennaRNA-backed ScanFold output.  Use "
              "--params <rna_turner2004.par> (or params.load_par) for published parameters." % paramset.source,
              file=stream or sys.stderr)
    return is_reconstructed(paramset)


_default = None


def default_params():
    global _default
    if _default is None:
        _default = load_par(DEFAULT_PAR)
    return _default.copy()


def random_params(seed, scale=200):
    """A randomised but symmetry-respecting set for index-order tests (oracle vs HIP vs evaluator).

    Keeps loop-initiation arrays / constants of the default set and replaces every sequence-dependent
    table by random integers, honouring stack[a][b]==stack[b][a],
    int11[a][b][x][y]==int11[b][a][y][x], int22[a][b][w][x][y][z]==int22[b][a][y][z][w][x]
     — the symmetries a loop read from either strand relies on.
    """
    rng = np.random.default_rng(seed)
    p = default_params()
    r = p.rec

    def rnd(shape, lo=-scale, hi=scale):
        return rng.integers(lo, hi, size=shape)

    st = rnd((8, 8), -350, 50)
    st = np.minimum(st, st.T)
    r["stack"] = st
    for f in ("mismatchI", "mismatchH", "mismatchM", "mismatch1nI", "mismatch23I", "mismatchExt"):
        r[f] = rnd((8, 5, 5), -250, 60)
    r["dangle5"] = rnd((8, 5), -120, 0)
    r["dangle3"] = rnd((8, 5), -180, 0)
    a = rnd((8, 8, 5, 5), -100, 300)
    r["int11"] = np.minimum(a, a.transpose(1, 0, 3, 2))
    r["int21"] = rnd((8, 8, 5, 5, 5), 50, 500)
    b = rnd((8, 8, 5, 5, 5, 5), -50, 400)
    r["int22"] = np.minimum(b, b.transpose(1, 0, 4, 5, 2, 3))
    for f in ("tetra_E", "tri_E", "hexa_E"):
        r[f] = rnd((MAX_SPECIAL,), 100, 700)
    r["TerminalAU"] = int(rng.integers(20, 90))
    r["MLbase"] = int(rng.integers(0, 3)) * 10
    r["MLintern"][:] = int(rng.integers(-120, -30))
    r["MLclosing"] = int(rng.integers(300, 1100))
    p.source = "random_params(%d)" % seed
    return p

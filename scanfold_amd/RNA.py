"""A minimal `RNA`-shaped facade over the HIP engine — just the names ScanFold calls.

reference call sites: RNA.md() ScanFold-Scan.py:70, ScanFold.py:212; RNA.fold_compound(seq, md) Scan:382;
fc.mfe() :385; fc.pf() :383; RNA.pf_fold(seq) :384; fc.centroid() :388; fc.mean_bp_distance() :389;
RNA.fold(seq) :245.  Energies come back as ViennaRNA returns them: (float)dcal / 100 widened to a Python float.
md.max_bp_span (ScanFold.py:214-215) is honoured (sf_set_max_bp_span).
md.temperature other than 37 needs a parameter set with enthalpy tables (params.load_par of a real .par file).
fc.hc_add_from_db and fc.sc_add_SHAPE_deigan (ScanFold-Scan.py:410; ScanFold.py:512,534) are provided through
sf_fold_constrained.  Not provided: duplexfold, plotting, Zarringhalam soft constraints (upstream's call fails too).
"""
import numpy as np

from . import _lib


class md:
    def __init__(self):
        self.temperature = 37.0
        self.max_bp_span = -1
        self.dangles = 2


def _check_md(model):
    eng = _lib.get_engine()
    eng.set_temperature(float(model.temperature) if model is not None else 37.0)
    span = getattr(model, "max_bp_span", -1) if model is not None else -1
    span = int(span) if span not in (None, -1) and int(span) > 0 else 0
    if span != getattr(eng, "_span", 0):  # the model is global state of the engine, like RNA's md defaults
        eng.set_max_bp_span(span)
        eng._span = span
    return eng


def _f32(dcal):
    return float(np.float32(dcal) / np.float32(100.0))


def deigan_pseudo_energies(reactivities, m, b, n):
    """vrna_sc_add_SHAPE_deigan [EXT]: position i (1-based) gets m*ln(reactivity[i] + 1) + b kcal/mol, 0 where the
    reactivity is negative (missing data); stored as (int)roundf(x * 100) dcal/mol.  The C function reads the vector
    1-BASED (index 0 is a dummy, as in ViennaRNA's own Python test the reference cites, ScanFold.py:219-221), and
    ScanFold passes a 0-based window slice (ScanFold.py:523,534), so upstream applies reactivity k+1 to nucleotide k and
    reads one element past the end for the last nucleotide.  That shift is reproduced here; the out-of-range element
    counts as missing data.  -> int32 array of n pseudo-energies (0-based)."""
    import math
    vals = [float(v) for v in reactivities]
    out = np.zeros(n, dtype=np.int32)
    for i in range(1, n + 1):
        r = vals[i] if i < len(vals) else -999.0
        x = 0.0 if r < 0 else m * math.log(r + 1.0) + b
        v = float(np.float32(x * 100.0))
        out[i - 1] = int(math.copysign(math.floor(abs(v) + 0.5), v))  # roundf: halves away from zero
    return out


class fold_compound:
    def __init__(self, sequence, model=None):
        self.sequence = str(sequence)
        self._model = model
        self._eng = _check_md(model)
        self._pf = None
        self._hc = None   # dot-bracket hard constraint (hc_add_from_db)
        self._sc = None   # Deigan pseudo-energies, int32 dcal per nucleotide (sc_add_SHAPE_deigan)

    def hc_add_from_db(self, constraint, options=None):
        """fc.hc_add_from_db(window_constraints) (ScanFold-Scan.py:410; ScanFold.py:512): '.', 'x', '|', '<', '>', '(' ')'
        with ViennaRNA's default (non-enforcing) meaning, see include/scanfold_hip.h: sf_fold_constrained."""
        constraint = str(constraint)
        if len(constraint) != len(self.sequence):
            raise ValueError("constraint string and sequence differ in length")
        self._hc = constraint
        self._pf = None
        return 1

    def sc_add_SHAPE_deigan(self, reactivities, m, b, options=None):
        """fc.sc_add_SHAPE_deigan(window_react_list, slope, intercept) (ScanFold.py:534,539)."""
        self._sc = deigan_pseudo_energies(reactivities, float(m), float(b), len(self.sequence))
        return 1

    def sc_add_SHAPE_zarringhalam(self, reactivities, b=None, default_value=None, shape_conversion=None, options=None):
        """ScanFold.py:536 calls this with ONE argument; ViennaRNA's binding needs b, default_value and the conversion
        string as well, so upstream's --shapeZ path ends in a TypeError.  Same here."""
        if b is None or default_value is None or shape_conversion is None:
            raise TypeError("sc_add_SHAPE_zarringhalam() missing required arguments: b, default_value, shape_conversion")
        raise NotImplementedError("Zarringhalam soft constraints are not implemented by the HIP engine")

    def mfe(self):
        self._eng = _check_md(self._model)
        if self._hc is not None or self._sc is not None:
            r = self._eng.fold_constrained([self.sequence], None if self._hc is None else [self._hc],
                                           None if self._sc is None else self._sc[None, :], pf=False)
            return r["structure"][0], _f32(r["mfe"][0])
        e, db = self._eng.mfe_trace_batch([self.sequence])
        return db[0], _f32(e[0])

    def pf(self):
        """-> (structure string, ensemble free energy).  The string is the centroid structure, not
        ViennaRNA's pair-propensity string (ScanFold discards it, ScanFold-Scan.py:383)."""
        self._eng = _check_md(self._model)
        if self._sc is not None:
            raise NotImplementedError("partition function with SHAPE soft constraints (the reference calls fc.pf() "
                                      "before sc_add_SHAPE_*, ScanFold.py:525-539)")
        if self._hc is not None:
            r = self._eng.fold_constrained([self.sequence], [self._hc], None, mfe=False)
        else:
            r = self._eng.pf_batch([self.sequence])
        self._pf = r
        return r["centroid"][0], float(r["dG"][0])

    def centroid(self):
        if self._pf is None:
            self.pf()
        return self._pf["centroid"][0], float(self._pf["centroid_dist"][0])

    def mean_bp_distance(self):
        if self._pf is None:
            self.pf()
        return float(self._pf["mean_bp_dist"][0])


def fold(sequence):
    return fold_compound(sequence).mfe()


def pf_fold(sequence):
    return fold_compound(sequence).pf()

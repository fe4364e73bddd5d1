"""A minimal `RNA`-shaped facade over the HIP engine — just the names ScanFold calls.

reference call sites: RNA.md() ScanFold-Scan.py:70, ScanFold.py:212; RNA.fold_compound(seq, md) Scan:382;
fc.mfe() :385; fc.pf() :383; RNA.pf_fold(seq) :384; fc.centroid() :388; fc.mean_bp_distance() :389;
RNA.fold(seq) :245.  Energies come back as ViennaRNA returns them: (float)dcal / 100 widened to a Python float.
md.max_bp_span (ScanFold.py:214-215) is honoured (sf_set_max_bp_span).
Not provided (out of this path's scope): hard/soft constraints, duplexfold, plotting.
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
    if model is not None:
        if float(model.temperature) != eng.params.temperature:
            raise NotImplementedError("temperature %s C: parameter set valid at %s C only"
                                      % (model.temperature, eng.params.temperature))
    span = getattr(model, "max_bp_span", -1) if model is not None else -1
    span = int(span) if span not in (None, -1) and int(span) > 0 else 0
    if span != getattr(eng, "_span", 0):  # the model is global state of the engine, like RNA's md defaults
        eng.set_max_bp_span(span)
        eng._span = span
    return eng


def _f32(dcal):
    return float(np.float32(dcal) / np.float32(100.0))


class fold_compound:
    def __init__(self, sequence, model=None):
        self.sequence = str(sequence)
        self._eng = _check_md(model)
        self._pf = None

    def mfe(self):
        e, db = self._eng.mfe_trace_batch([self.sequence])
        return db[0], _f32(e[0])

    def pf(self):
        """-> (structure string, ensemble free energy).  The string is the centroid structure, not
        ViennaRNA's pair-propensity string (ScanFold discards it, ScanFold-Scan.py:383)."""
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

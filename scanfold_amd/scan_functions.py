"""The helper functions ScanFold-Scan.py defines inline (they differ from ScanFoldFunctions' versions).

reference                                      here
---------------------------------------------  -----------------------------------------------
pscore_function(energy_list, r)   Scan:218-229  same
zscore_function(energy_list, r)   Scan:232-242  np.std variant; the STRING "#DIV/0!" when sd == 0
rna_folder(frag)                  Scan:244-246  global 37 C model, whatever -t says
energies(seq_list)                Scan:253-262  one batched device launch
scramble / dinuclShuffle / randomizer           shared with scanfold_amd.functions (identical upstream,
                                                 except that Scan's scramble does not transcribe)
(Scan = /root/reference/ScanFold-Scan.py)
"""
import numpy as np

from . import functions as _sff
from .functions import dinuclShuffle, multiprocessing, randomizer  # noqa: F401  (same names upstream)


def pscore_function(energy_list, randomizations):
    """Fraction of all r+1 energies below the native one (ScanFold-Scan.py:218-229)."""
    below, total = _sff.count_below_native(energy_list)
    return float(below) / float(total)


def zscore_function(energy_list, randomizations):
    """ScanFold-Scan.py:232-242: population standard deviation (np.std) over native + shuffles, np.mean over
    energy_list[1:randomizations] (the last shuffle left out), the STRING "#DIV/0!" for a flat list.  One row of the
    vectorised form below — proven bit-equal to the per-list numpy calls in tests/test_golden_host.py."""
    z, flat = zscores_rows(np.asarray([list(energy_list)], dtype=np.float64), randomizations)
    return "#DIV/0!" if flat[0] else z[0]


def rna_folder(frag):
    return _sff.energies([str(frag)], 37, "rnafold")[0]


def energies(seq_list):
    return _sff.energies([str(s) for s in seq_list], 37, "rnafold")


def scramble(text, randomizations, type):
    """As ScanFoldFunctions.scramble but WITHOUT the T -> U step (ScanFold-Scan.py:266-282): the caller transcribes."""
    window = str(text)
    if type == "di":
        return [dinuclShuffle(window) for _ in range(randomizations)]
    if type == "mono":
        return [randomizer(window) for _ in range(randomizations)]
    print(_sff._SHUFFLE_MESSAGE)
    return []


# ---- row-vectorised forms used by the scan driver; bit-equal to the per-row functions above ----
def zscores_rows(E, randomizations):
    """E: float64 (n, r+1).  Returns (z float64 (n,), sd_is_zero bool (n,)); z is undefined where sd == 0."""
    E = np.asarray(E, dtype=np.float64)
    sd = E.std(axis=1)
    mean = E[:, 1:randomizations].mean(axis=1) if randomizations > 1 else np.full(len(E), np.nan)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = (E[:, 0] - mean) / sd
    return z, sd == 0


def pscores_rows(E):
    E = np.asarray(E, dtype=np.float64)
    return (E < E[:, :1]).sum(axis=1) / float(E.shape[1])

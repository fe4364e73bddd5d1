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
    below_native = 0
    total_count = len(energy_list)
    native_mfe = float(energy_list[0])
    for MFE in energy_list:
        if float(MFE) < float(native_mfe):
            below_native += 1
    return float(float(below_native) / float(total_count))


def zscore_function(energy_list, randomizations):
    sd = np.std(energy_list)
    native_mfe = energy_list[0]
    scrambled_mean_mfe = np.mean(energy_list[1:randomizations])
    if sd != 0:
        zscore = (native_mfe - scrambled_mean_mfe) / sd
    if sd == 0:
        zscore = "#DIV/0!"
    return zscore


def rna_folder(frag):
    return _sff.energies([str(frag)], 37, "rnafold")[0]


def energies(seq_list):
    return _sff.energies([str(s) for s in seq_list], 37, "rnafold")


def scramble(text, randomizations, type):
    frag = str(text)
    frag_seqs = []
    if type == "di":
        for _ in range(randomizations):
            frag_seqs.append(dinuclShuffle(frag))
    elif type == "mono":
        frag_seqs = [randomizer(frag) for _ in range(randomizations)]
    else:
        print("Shuffle type not properly designated; please input \"di\" or \"mono\"")
    return frag_seqs


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

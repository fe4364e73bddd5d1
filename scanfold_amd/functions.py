"""Host-side mirror of the `ScanFoldFunctions` names that sit on ScanFold's scan hot path.

Same names, argument meaning and error behaviour as the reference module, so ScanFold's scan loop
(`ScanFold.py:420-757`) runs unchanged on top of it; the arithmetic underneath is the HIP engine
(`scanfold_amd._lib`), never ViennaRNA and never a CPU fold.

reference                                            here
---------------------------------------------------  ---------------------------------------------------
multiprocessing(func, args, workers)   SFF:140-144   same signature; no process pool is created
computeCountAndLists/chooseEdge/…      SFF:155-254   folded into dinuclShuffle (same RNG consumption)
dinuclShuffle(s)                       SFF:255-277   identical strings under the same random.seed()
simple_transcribe(seq)                 SFF:580-584
pvalue_function(energy_list, r)        SFF:727-738
zscore_function(energy_list, r)        SFF:741-751   statistics.stdev variant, 0.0 when sd == 0
rna_folder((frag, T, algo))            SFF:774-789   one batched device launch of size 1
randomizer(frag)                       SFF:800-802
energies(seq_list, T, algo)            SFF:805-814   ONE batched launch for the whole list
scramble(text, r, type)                SFF:834-851
get_gc_content(frag)                   SFF:1023-1036
get_dinucleotide_counts(frag)          SFF:1079-1094
(SFF = /root/reference/ScanFoldFunctions.py)
"""
import random
import statistics

import numpy as np

_ALPHABET = "ACGU"


def multiprocessing(func, args, workers):
    """Reference: a new 12-process pool per call.  Here: the batched device path for the fold
    functions, a plain in-process map for anything else.  `workers` is accepted and ignored."""
    args = list(args)
    if func is rna_folder and args:
        first = args[0]
        if isinstance(first, tuple):
            temps = {int(a[1]) for a in args}
            algos = {a[2] for a in args}
            if len(temps) == 1 and len(algos) == 1:
                return energies([a[0] for a in args], temps.pop(), algos.pop())
        else:
            return energies(args)
    return [func(a) for a in args]


# ----------------------------------------------------------------------------- shuffles
def dinuclShuffle(s):
    """Altschul-Erikson dinucleotide shuffle exactly as the reference performs it (P. Clote's 2003
    formulation): one random.random() per vertex to pick its last edge (retry until every vertex
    reaches the last character), then int(random.random()*barrier) swaps per successor list.
    Under the same `random` state it returns the same string as the reference function."""
    s = s.upper().replace("T", "U")
    n = len(s)
    idx = {"A": 0, "C": 1, "G": 2, "U": 3}
    codes = [idx[ch] for ch in s]  # KeyError on anything else, as in the reference
    present = [x for x in range(4) if x in codes]
    last = codes[-1]
    cnt = [[0] * 4 for _ in range(4)]
    succ = [[] for _ in range(4)]
    for a, b in zip(codes, codes[1:]):
        cnt[a][b] += 1
        succ[a].append(b)
    while True:
        last_edge = {}
        for x in present:
            if x == last:
                continue
            z = random.random()
            row = cnt[x]
            denom = float(row[0] + row[1] + row[2] + row[3])
            acc = 0
            pick = 3
            for y in range(3):
                acc += row[y]
                if z < float(acc) / denom:
                    pick = y
                    break
            last_edge[x] = pick
        reach = {x: 0 for x in present}
        for a, b in last_edge.items():
            if b == last:
                reach[a] = 1
        for _ in range(2):
            for a, b in last_edge.items():
                if reach[b] == 1:
                    reach[a] = 1
        if all(reach[x] for x in present if x != last):
            break
    for a, b in last_edge.items():
        succ[a].remove(b)
    for x in present:
        L = succ[x]
        barrier = len(L)
        for _ in range(len(L) - 1):
            z = int(random.random() * barrier)
            L[z], L[barrier - 1] = L[barrier - 1], L[z]
            barrier -= 1
    for a, b in last_edge.items():
        succ[a].append(b)
    out = [codes[0]]
    ptr = [0, 0, 0, 0]
    prev = codes[0]
    for _ in range(n - 2):
        ch = succ[prev][ptr[prev]]
        ptr[prev] += 1
        out.append(ch)
        prev = ch
    out.append(codes[-1])
    return "".join(_ALPHABET[c] for c in out)


def randomizer(frag):
    return "".join(random.sample(frag, len(frag)))


def simple_transcribe(seq):
    """T -> U.  (Upstream returns from inside a loop over the characters, so an empty string gives None.)"""
    return seq.replace("T", "U") if seq else None


_SHUFFLE_MESSAGE = "Shuffle type not properly designated; please input \"di\" or \"mono\""


def scramble(text, randomizations, type):
    """`randomizations` shuffled copies (ScanFoldFunctions.py:834-851): "di" transcribes first and keeps the dinucleotide
    counts, "mono" permutes; anything else prints upstream's message and yields no shuffles.  One generator draw
    sequence per copy, in order — the strings under random.seed() are upstream's (tests/test_golden_host.py)."""
    window = str(text)
    if type == "di":
        window = simple_transcribe(window)
        return [dinuclShuffle(window) for _ in range(randomizations)]
    if type == "mono":
        return [randomizer(window) for _ in range(randomizations)]
    print(_SHUFFLE_MESSAGE)
    return []


# ----------------------------------------------------------------------------- folds
def _dcal_to_float(dcal):
    """ViennaRNA returns (float)energy/100. — a C float widened to a Python float."""
    return [float(v) for v in (np.asarray(dcal, dtype=np.float32) / np.float32(100.0))]


def energies(seq_list, temperature=37, algo="rnafold"):
    """MFE (kcal/mol) of every sequence, order preserved.  One device launch per distinct length."""
    from . import _lib
    if algo != "rnafold":
        # the reference leaves MFE unbound for any other algo (ScanFoldFunctions.py:785-789)
        raise UnboundLocalError("local variable 'MFE' referenced before assignment")
    eng = _lib.get_engine()
    eng.set_temperature(int(temperature))  # md.temperature = int(temperature), ScanFoldFunctions.py:776-777
    seqs = [str(s) for s in seq_list]
    out = [None] * len(seqs)
    by_len = {}
    for k, s in enumerate(seqs):
        by_len.setdefault(len(s), []).append(k)
    for n, ks in by_len.items():
        if n == 0:
            for k in ks:
                out[k] = 0.0
            continue
        vals = _dcal_to_float(eng.mfe_batch([seqs[k] for k in ks]))
        for k, v in zip(ks, vals):
            out[k] = v
    return out


def rna_folder(arg):
    if isinstance(arg, tuple):
        frag, temperature, algo = arg
        return energies([frag], temperature, algo)[0]
    return energies([arg])[0]


# ----------------------------------------------------------------------------- statistics
def count_below_native(energy_list):
    """How many entries of the list are strictly below its first one (the native fold is never below itself)."""
    values = [float(e) for e in energy_list]
    return sum(1 for e in values if e < values[0]), len(values)


def pvalue_function(energy_list, randomizations):
    """Fraction of ALL r+1 energies below the native one (ScanFoldFunctions.py:727-738); `randomizations` is unused upstream too."""
    below, total = count_below_native(energy_list)
    return float(below) / float(total)


def zscore_function(energy_list, randomizations):
    """(native - mean of the shuffles) / sample standard deviation (ScanFoldFunctions.py:741-751).  The quirks are
    upstream's: the mean runs over energy_list[1:randomizations], which leaves the LAST shuffle out, while the deviation
    (n-1 denominator, exact `statistics` arithmetic) covers native and all shuffles; a flat list gives 0.0."""
    spread = statistics.stdev(energy_list)
    background = statistics.mean(energy_list[1:randomizations])
    if spread == 0:
        return float(0)
    return (energy_list[0] - background) / spread


# ----------------------------------------------------------------------------- composition helpers
_DINUCLEOTIDE_ORDER = tuple(a + b for a in "AUGC" for b in "AUGC")


def get_gc_content(frag):
    """(G + C) / (A + T/U + G + C), either case, 5 decimals (ScanFoldFunctions.py:1023-1036).  Upstream guards the
    division with `'C' and 'G' in frag`, which Python reads as `'G' in frag`: a window without an upper-case G gets 0."""
    text = str(frag)
    if "G" not in text:
        return 0
    n = {base: text.count(base) + text.count(base.lower()) for base in "ACGTU"}
    strong = n["G"] + n["C"]
    return round(float(strong) / float(n["A"] + n["T"] + n["U"] + strong), 5)


def get_dinucleotide_counts(frag):
    """Occurrences of the 16 dinucleotides at every offset, in upstream's order AA AU AG AC UA .. CC
    (ScanFoldFunctions.py:1079-1094)."""
    text = str(frag)
    seen = {}
    for k in range(len(text) - 1):
        seen[text[k:k + 2]] = seen.get(text[k:k + 2], 0) + 1
    return [seen.get(d, 0) for d in _DINUCLEOTIDE_ORDER]

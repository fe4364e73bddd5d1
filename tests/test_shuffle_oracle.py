"""The shuffle background without a GPU.

Three things are pinned here:
  1. the REFERENCE's dinuclShuffle (ScanFoldFunctions.py:255-277), run 30 000 times per input by
     tests/golden/make_golden_shuffle_hist.py, is uniform over the exhaustively enumerated set of sequences with the
     same dinucleotide counts and end characters — that is the distribution any replacement has to reproduce;
  2. the host twin (scanfold_amd/functions.py) gives the same histogram, draw for draw, under the same seed;
  3. oracle/sf_shuffle_oracle.c — the reference's algorithm on the product's Philox stream — has that
     distribution too, keeps the invariants on windows with N, and is bit-equal to the device kernel's logic
     (the kernel source compiled for the CPU, tests/emul).  On the GPU the same comparison runs against the
     real kernel (tests/test_gpu_parity.py).
"""
import json
import os
import random
from collections import Counter

import numpy as np
import pytest

from shuffle_util import chi2_limit, chi2_two_sample, chi2_uniform, codes_to_str, di_arrangements

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref_hist():
    with open(os.path.join(ROOT, "tests", "golden", "reference_dinucl_hist.json")) as f:
        return json.load(f)["items"]


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def test_reference_dinuclshuffle_is_uniform_over_the_enumerated_set(ref_hist):
    for it in ref_hist:
        allowed = di_arrangements(it["s"])
        assert sorted(it["hist"]) == allowed, it["s"]  # every arrangement occurs, nothing else does
        assert sum(it["hist"].values()) == it["draws"]
        assert chi2_uniform(list(it["hist"].values())) < chi2_limit(len(allowed) - 1), it["s"]


def test_host_twin_reproduces_the_reference_histogram_draw_for_draw(ref_hist):
    from scanfold_amd import functions as sff
    for it in ref_hist[:3]:
        random.seed(it["seed"])
        hist = Counter(sff.dinuclShuffle(it["s"]) for _ in range(it["draws"]))
        assert dict(hist) == it["hist"], it["s"]


def test_oracle_di_shuffle_has_the_reference_distribution(ref_hist, orc):
    for k, it in enumerate(ref_hist):
        s = it["s"]
        allowed = di_arrangements(s)
        rows = codes_to_str(orc.shuffle_windows(s, len(s), 1, 0, 1, 40000, 1, 77 + k))
        assert rows[0] == s
        hist = Counter(rows[1:])
        assert sorted(hist) == allowed, s
        assert chi2_uniform([hist[a] for a in allowed]) < chi2_limit(len(allowed) - 1), s
        # and it is statistically indistinguishable from the reference's own draws
        assert chi2_two_sample([hist[a] for a in allowed], [it["hist"][a] for a in allowed]) < chi2_limit(
            len(allowed) - 1), s


def test_oracle_mono_shuffle_is_a_uniform_permutation(orc):
    s = "ACGUA"
    rows = codes_to_str(orc.shuffle_windows(s, 5, 1, 0, 1, 60000, 0, 5))[1:]
    hist = Counter(rows)
    assert len(hist) == 60  # 5! / 2! distinct strings
    assert all(Counter(r) == Counter(s) for r in hist)
    assert chi2_uniform(list(hist.values())) < chi2_limit(59)


def test_oracle_shuffle_invariants_with_n_and_window_independence(orc):
    rng = np.random.default_rng(3)
    tr = "".join("ACGUN"[k] for k in rng.choice(5, 900, p=[0.24, 0.24, 0.24, 0.24, 0.04]))
    W, step, r = 60, 7, 6
    nwin = (len(tr) - W) // step + 1
    for kind in (0, 1):
        rows = codes_to_str(orc.shuffle_windows(tr, W, step, 0, nwin, r, kind, 9))
        for w in range(nwin):
            nat = tr[w * step:w * step + W]
            assert rows[w * (r + 1)] == nat
            for k in range(1, r + 1):
                s = rows[w * (r + 1) + k]
                if kind == 0:
                    assert Counter(s) == Counter(nat)
                else:
                    assert (s[0], s[-1]) == (nat[0], nat[-1])
                    assert Counter(zip(s, s[1:])) == Counter(zip(nat, nat[1:]))
        # a window's shuffles depend on (seed, absolute window index, k) only: any sub-range reproduces them
        part = codes_to_str(orc.shuffle_windows(tr, W, step, 11, 4, r, kind, 9))
        assert part == rows[11 * (r + 1):15 * (r + 1)]


@pytest.fixture(scope="module")
def emul():
    from emul_engine import emul_engine
    return emul_engine()


def test_device_kernel_logic_equals_the_oracle_bit_for_bit(emul, orc):
    """sf_shuffle_kernel (compiled for the CPU) vs sf_shuffle_oracle.c: cfg1's windows and odd shapes incl. N."""
    seq = "".join("ACGU"[k] for k in np.random.default_rng(1).integers(0, 4, 1000))  # BASELINE config 1
    for kind in (0, 1):
        assert (emul.shuffle_windows(seq, 120, 40, 0, 23, 10, kind, 3) ==
                orc.shuffle_windows(seq, 120, 40, 0, 23, 10, kind, 3)).all()
    rng = np.random.default_rng(2)
    tr = "".join("ACGUNt"[k] for k in rng.choice(6, 700, p=[0.23, 0.23, 0.23, 0.23, 0.05, 0.03]))
    for kind in (0, 1):
        for (W, step, wb, nw, r, seed) in [(120, 7, 0, 30, 9, 5), (10, 1, 3, 300, 17, 2 ** 40 + 3), (1, 1, 0, 5, 3, 1),
                                           (2, 1, 0, 5, 3, 1), (37, 11, 2, 20, 5, 99), (200, 3, 0, 40, 4, 12)]:
            assert (emul.shuffle_windows(tr, W, step, wb, nw, r, kind, seed) ==
                    orc.shuffle_windows(tr, W, step, wb, nw, r, kind, seed)).all(), (kind, W)

"""Test helpers for the shuffle background: exhaustive enumeration of what a dinucleotide shuffle may return."""
from collections import Counter

import numpy as np


def di_arrangements(s):
    """Every sequence with the dinucleotide counts, first and last character of `s` (sorted list)."""
    target = Counter(zip(s, s[1:]))
    n = len(s)
    alphabet = sorted(set(s))
    out = []

    def rec(cur, left):
        if len(cur) == n - 1:
            if left.get((cur[-1], s[-1]), 0) == 1:
                out.append(cur + s[-1])
            return
        for ch in alphabet:
            k = (cur[-1], ch)
            if left.get(k, 0) > 0:
                left[k] -= 1
                rec(cur + ch, left)
                left[k] += 1

    rec(s[0], dict(target))
    return sorted(set(out))


def chi2_uniform(counts):
    counts = np.asarray(counts, dtype=np.float64)
    e = counts.sum() / len(counts)
    return float(((counts - e) ** 2 / e).sum())


def chi2_limit(dof, p=0.999):
    from scipy.stats import chi2
    return float(chi2.ppf(p, dof))


def chi2_two_sample(a, b):
    """Two histograms over the same categories (possibly different totals): Pearson statistic, dof = k - 1."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ka, kb = np.sqrt(b.sum() / a.sum()), np.sqrt(a.sum() / b.sum())
    return float(((ka * a - kb * b) ** 2 / (a + b)).sum())


def codes_to_str(rows):
    return [bytes(r).decode() for r in np.frombuffer(b"NACGU", dtype=np.uint8)[rows]]

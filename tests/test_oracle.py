"""What pins the oracle in the absence of any ViennaRNA golden value."""
import math
import re
from functools import lru_cache

import numpy as np
import pytest

from scanfold_amd import params

KT = (37 + 273.15) * 1.98717


def rseq(rng, n, p=None):
    return "".join("ACGU"[k] for k in rng.choice(4, n, p=p))


def count_structures(s):
    ok = lambda a, b: a + b in ("CG", "GC", "GU", "UG", "AU", "UA")

    @lru_cache(None)
    def N(i, j):
        if j - i < 4:
            return 1
        t = N(i + 1, j)
        for k in range(i + 4, j + 1):
            if ok(s[i], s[k]):
                t += N(i + 1, k - 1) * N(k + 1, j)
        return t
    return N(0, len(s) - 1)


def test_v1_bruteforce_equals_dp_and_traceback_evaluates_to_mfe(oracle):
    rng = np.random.default_rng(0)
    for t in range(150):
        s = rseq(rng, int(rng.integers(8, 20)), [0.15, 0.35, 0.35, 0.15] if t % 2 else None)
        db, e = oracle.mfe(s)
        be, Z, _, cnt = oracle.brute(s)
        assert cnt == count_structures(s)
        assert e == be == oracle.eval_structure(s, db), (s, db, e, be)


def test_v1_v3_with_randomised_parameter_tables():
    from oracle import oracle as orc
    rng = np.random.default_rng(1)
    n_multi = 0
    for seed in range(4):
        p = params.random_params(seed)
        if seed % 2:
            p.rec["MLclosing"] = -200  # make multiloops win
        orc.set_params(p)
        for t in range(25):
            s = rseq(rng, int(rng.integers(10, 21)), [0.15, 0.35, 0.35, 0.15] if t % 2 else None)
            db, e = orc.mfe(s)
            be, Z, bpp, _ = orc.brute(s, True)
            assert e == be == orc.eval_structure(s, db), (seed, s, db)
            r = orc.pf(s, True)
            assert abs(-math.log(Z) * KT / 1000 - r["dG"]) < 1e-9
            assert np.abs(bpp - r["bpp"]).max() < 1e-9
            n_multi += bool(re.search(r"\([^()]*\([^()]*\)[^()]*\(", db))
    assert n_multi > 0


def test_v2_traceback_energy_on_long_sequences(oracle):
    rng = np.random.default_rng(2)
    for n in (30, 60, 120, 200):
        for _ in range(10):
            s = rseq(rng, n)
            db, e = oracle.mfe(s)
            assert len(db) == n and db.count("(") == db.count(")")
            assert oracle.eval_structure(s, db) == e
    assert oracle.mfe("A" * 50) == ("." * 50, 0)
    assert oracle.mfe("N" * 50)[1] == 0
    assert oracle.mfe("ACG")[1] == 0


def test_v3_partition_function_against_enumeration(oracle):
    rng = np.random.default_rng(3)
    for _ in range(40):
        s = rseq(rng, int(rng.integers(8, 18)))
        be, Z, bpp, _ = oracle.brute(s, True)
        r = oracle.pf(s, True)
        assert abs(-math.log(Z) * KT / 1000 - r["dG"]) < 1e-9
        assert np.abs(bpp - r["bpp"]).max() < 1e-9
        assert (r["bpp"].sum(axis=1) <= 1 + 1e-9).all()
        assert abs(r["mean_bp_dist"] - 2 * (bpp * (1 - bpp)).sum()) < 1e-9
        cen = ["."] * len(s)
        for i, j in zip(*np.where(bpp > 0.5)):
            cen[i - 1], cen[j - 1] = "(", ")"
        assert "".join(cen) == r["centroid"]
        assert r["dG"] <= be / 100.0 + 1e-9  # ensemble free energy is below the MFE


def test_case_and_t_are_normalised(oracle):
    assert oracle.mfe("GGGGAAAACCCC") == oracle.mfe("ggggaaaacccc") == oracle.mfe("GGGGAAAACCCC".replace("U", "T"))
    assert oracle.mfe("GGGGUUUUCCCC") == oracle.mfe("GGGGTTTTCCCC")


def test_known_small_answers(oracle):
    # hand-evaluated with the shipped table: GGGGAAAACCCC = 3 stacks + 4-nt hairpin with GA.. mismatch
    p = params.default_params().rec
    db, e = oracle.mfe("GGGGAAAACCCC")
    assert db == "((((....))))"
    # G-C closing pairs are type 2; the pair stacked inside is seen reversed (type 1); the exterior stem of a
    # sequence-spanning helix has no neighbours and GC carries no TerminalAU
    exp = 3 * p["stack"][2][1] + p["hairpin"][4] + p["mismatchH"][2][1][1]
    assert e == exp == -540
    # special tetraloop replaces initiation + mismatch outright
    db, e = oracle.mfe("GGGGCUUCGGCCCC")
    assert db == "(((((....)))))" or db.count("(") >= 4
    assert oracle.eval_structure("GGGGCUUCGGCCCC", db) == e


def test_max_bp_span_dp_equals_enumeration(oracle):
    """RNA.md().max_bp_span (ScanFold.py:214-215): with the limit set, the DP, the exhaustive enumeration and the
    partition function agree again (the enumeration only builds structures whose pairs respect the span)."""
    import numpy as np
    rng = np.random.default_rng(77)
    try:
        for span in (6, 9, 12):
            oracle.set_max_bp_span(span)
            for _ in range(6):
                s = "".join("ACGU"[k] for k in rng.integers(0, 4, 15))
                e, Z, bpp, cnt = oracle.brute(s, want_bpp=True)
                db, e_dp = oracle.mfe(s)
                assert e_dp == e and oracle.eval_structure(s, db) == e
                stack = []
                for k, ch in enumerate(db):
                    if ch == "(":
                        stack.append(k)
                    elif ch == ")":
                        assert k - stack.pop() + 1 <= span
                r = oracle.pf(s, want_bpp=True)
                assert abs(np.exp(-r["dG"] * 1000.0 / (1.98717 * 310.15)) - Z) <= 1e-9 * Z
                assert np.abs(r["bpp"] - bpp).max() < 1e-9
    finally:
        oracle.set_max_bp_span(0)


def test_cpu_twin_equals_the_oracle(oracle):
    """oracle/sf_cpu_twin.c (bench.py's CPU baseline engine) against the checker: same energies, structures,
    centroids; ensemble diversity to 1e-9 — several widths, a window with N, one OpenMP thread and several."""
    import numpy as np
    rng = np.random.default_rng(2)
    for W, n, r, nt in ((16, 30, 4, 1), (30, 40, 5, 2), (61, 12, 3, 1), (120, 6, 8, 2), (200, 2, 2, 1)):
        rows = np.frombuffer(b"ACGUN", dtype=np.uint8)[rng.choice(5, (n * (r + 1), W), p=[.245, .245, .245, .245, .02])]
        a = oracle.scan_windows(rows, n, r, nthreads=nt)
        b = oracle.twin_scan_windows(rows, n, r, nthreads=nt)
        assert (a["energies"] == b["energies"]).all(), W
        assert a["structure"] == b["structure"] and a["centroid"] == b["centroid"], W
        assert np.abs(a["ens_div"] - b["ens_div"]).max() < 1e-9
    arr = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (200, 90))]
    assert (oracle.twin_mfe_batch(arr, 2) == oracle.mfe_batch(arr, 2)).all()


def test_native_build_of_the_cpu_engine_is_a_second_instance_with_the_same_results(oracle):
    """bench.py's cpu_baseline times two builds of the same sources: the portable one and `-O3 -march=native` made on the
    host that runs it (oracle.build_native: oracle/_native/, stamped with the CPU it was built for, never shipped).  The native
    build is a separate library instance with its own parameter tables and must give the portable build's results."""
    import os
    import numpy as np
    from scanfold_amd import params
    L = oracle.lib_native()
    assert L is not oracle.lib()
    here = os.path.dirname(os.path.abspath(oracle.__file__))
    stamp = open(os.path.join(here, "_native", "built_for.txt")).read()
    assert "-march=native" in stamp and oracle.NATIVE_FLAGS in stamp and stamp == oracle._cpu_stamp()
    oracle.set_params(params.default_params(), L=L)
    rng = np.random.default_rng(5)
    rows = np.frombuffer(b"ACGU", dtype=np.uint8)[rng.integers(0, 4, (4 * 6, 120))]
    a = oracle.twin_scan_windows(rows, 4, 5, nthreads=2)
    b = oracle.twin_scan_windows(rows, 4, 5, nthreads=2, L=L)
    assert (a["energies"] == b["energies"]).all() and a["structure"] == b["structure"] and a["centroid"] == b["centroid"]
    assert np.abs(a["ens_div"] - b["ens_div"]).max() < 1e-9
    # the library files of .gitignore / .gpurunignore: the native object must not travel
    root = os.path.dirname(here)
    assert "oracle/_native/" in open(os.path.join(root, ".gitignore")).read()
    assert "oracle/_native/" in open(os.path.join(root, ".gpurunignore")).read()

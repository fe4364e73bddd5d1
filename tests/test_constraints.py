"""SURVEY.md §8(f) rank 1 and 3 without a GPU: hard constraints per window (`-c`, fc.hc_add_from_db), Deigan SHAPE
pseudo-energies (fc.sc_add_SHAPE_deigan) and folding temperatures other than 37 C.

reference: ScanFold-Scan.py:70-71 (md.temperature), :312-333 (constraint file), :405-418 (per-window slice,
hc_add_from_db, constrained mfe / pf / centroid / mean_bp_distance); ScanFold.py:218-262,522-544 (SHAPE).
The oracle gains the constraint in its single ptype() / stack term and is pinned the usual way — constrained DP ==
exhaustive enumeration of the admissible structures == independent evaluator, constrained partition function == brute
Boltzmann sum — and the general kernels (compiled for the CPU, tests/emul) must equal it.  ViennaRNA's meaning of the
constraint characters is [EXT] and unverifiable here (see oracle/sf_oracle.c)."""
import math

import numpy as np
import pytest

from scanfold_amd import _lib, params
from scanfold_amd import scan as scanmod


def random_constraint(rng, n, max_pairs=2):
    c = list(rng.choice(list("......x<>|"), n))
    for _ in range(int(rng.integers(0, max_pairs + 1))):
        a = int(rng.integers(0, n - 5))
        b = int(rng.integers(a + 4, n))
        if all(ch not in "()" for ch in c[a:b + 1]):
            c[a], c[b] = "(", ")"
    return "".join(c)


def canonical_constraint(rng, seq, max_pairs=3):
    """A constraint whose bracket pairs join bases that CAN pair (the LDS kernels take these; a bracket pair of
    non-complementary bases is a type-7 pair and goes to the general kernels)."""
    n = len(seq)
    c = list(rng.choice(list("......x<>|"), n))
    ok = {("G", "C"), ("C", "G"), ("G", "U"), ("U", "G"), ("A", "U"), ("U", "A")}
    for _ in range(max_pairs * 8):
        a = int(rng.integers(0, n - 5))
        b = int(rng.integers(a + 4, n))
        if (seq[a], seq[b]) in ok and all(ch not in "()" for ch in c[a:b + 1]) and c.count("(") < max_pairs:
            c[a], c[b] = "(", ")"
    return "".join(c)


def rseq(rng, n):
    return "".join("ACGU"[k] for k in rng.integers(0, 4, n))


def test_oracle_constrained_dp_equals_enumeration_and_evaluator(oracle):
    rng = np.random.default_rng(31)
    kT = (37 + 273.15) * 1.98717
    try:
        for t in range(250):
            n = int(rng.integers(8, 17))
            s = rseq(rng, n)
            cons = random_constraint(rng, n)
            sc = rng.integers(-80, 60, n).astype(np.int32) if t % 2 else None
            oracle.set_constraint(cons, sc)
            db, e = oracle.mfe(s)
            be, Z, _, cnt = oracle.brute(s)
            assert e == be == oracle.eval_structure(s, db), (s, cons)
            for k, ch in enumerate(cons):  # the structure obeys every character
                assert not (ch == "x" and db[k] != ".") and not (ch == "<" and db[k] == ")") and not (ch == ">" and db[k] == "(")
            stack = []
            for k, ch in enumerate(db):
                if ch == "(":
                    stack.append(k)
                elif ch == ")":
                    o = stack.pop()
                    if cons[o] == "(" or cons[k] == ")":  # bracket positions pair with their partner only
                        assert cons[o] == "(" and cons[k] == ")", (s, cons, db)
            if sc is None:  # the soft constraint is an MFE-only term (the reference adds it after fc.pf())
                assert abs(oracle.pf(s)["dG"] + math.log(Z) * kT / 1000) < 1e-9, (s, cons)
        # a forced non-complementary pair (type 7) that the MFE structure actually uses: A4-A9 inside a GC helix, made
        # attractive by a stacking bonus
        s, cons = "GGGAAAAAACCC", "...(....)..."
        sc = np.zeros(12, dtype=np.int32)
        sc[[2, 3, 8, 9]] = -300
        oracle.set_constraint(cons, sc)
        db, e = oracle.mfe(s)
        assert db[3] == "(" and db[8] == ")" and e == oracle.brute(s)[0] == oracle.eval_structure(s, db)
        oracle.set_constraint(None, sc)
        assert oracle.mfe(s)[0][3] != "("  # without the bracket pair A-A cannot pair
        oracle.set_constraint("((..", None)
        with pytest.raises(RuntimeError):
            oracle.mfe("ACGU")
    finally:
        oracle.set_constraint(None, None)


@pytest.fixture(scope="module")
def emul():
    from emul_engine import emul_engine
    return emul_engine()


def test_constrained_kernels_follow_the_oracle(emul, oracle):
    """sf_fold_constrained in both kernel modes: 0 = the LDS kernels with the constraint at their pair-type seam
    (sf_mfe_fast_kernel / sf_pf_lds_kernel, HC instantiations; batches with a bracket pair of non-complementary bases and
    folds that leave the int16 range fall back to the general kernels), 1 = the general int32 / FP64 kernels."""
    emul.load_params(params.default_params())
    rng = np.random.default_rng(5)
    try:
        for t in range(24):
            W = int(rng.choice([20, 33, 60, 120]))
            n = 2 if W == 120 else 3
            seqs = [rseq(rng, W) for _ in range(n)]
            # two batches out of three keep to pairs the LDS kernels' tables hold
            cons = [canonical_constraint(rng, s, 3) for s in seqs] if t % 3 else [random_constraint(rng, W, 3) for _ in range(n)]
            sc = rng.integers(-60, 40, (n, W)).astype(np.int32) if t % 2 else None
            emul.set_kernel_mode(1)
            r1 = emul.fold_constrained(seqs, cons, sc)
            emul.set_kernel_mode(0)
            r = emul.fold_constrained(seqs, cons, sc)
            assert r["structure"] == r1["structure"] and (r["mfe"] == r1["mfe"]).all() and r["centroid"] == r1["centroid"]
            assert np.abs(np.asarray(r["dG"]) - np.asarray(r1["dG"])).max() < 1e-9
            for k in range(n):
                oracle.set_constraint(cons[k], None if sc is None else sc[k])
                assert oracle.mfe(seqs[k]) == (r["structure"][k], int(r["mfe"][k])), (seqs[k], cons[k])
                oracle.set_constraint(cons[k], None)
                o = oracle.pf(seqs[k])
                assert o["centroid"] == r["centroid"][k]
                assert abs(o["dG"] - r["dG"][k]) < 1e-9 and abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < 1e-9
        # no constraint at all == the plain kernels; an all-dots constraint changes nothing either
        oracle.set_constraint(None, None)
        seqs = [rseq(rng, 50) for _ in range(4)]
        plain_e, plain_db = emul.mfe_trace_batch(seqs)
        for cons in (None, ["." * 50] * 4, ["|" * 50] * 4):
            r = emul.fold_constrained(seqs, cons)
            assert (r["mfe"] == plain_e).all() and r["structure"] == plain_db
        with pytest.raises(_lib.ScanFoldHipError, match="unbalanced"):
            emul.fold_constrained(["ACGUACGUACGUACGUACGU"], ["((.................."])
        assert emul.fold_constrained(["ACGU" * 5], ["." * 20])["mfe"][0] == oracle.mfe("ACGU" * 5)[1]  # flag was cleared
    finally:
        oracle.set_constraint(None, None)


def test_constrained_partition_function_above_120_uses_the_device_table_kernel(emul, oracle):
    """120 < W <= 250: the constrained partition function runs on sf_pf_fast_kernel's HC instantiation (kernel mode 0) —
    until round 4 it fell back to sf_pf_kernel, ~90 x slower per fold.  Both modes equal each other and the oracle."""
    emul.load_params(params.default_params())
    rng = np.random.default_rng(8)
    try:
        for W in (121, 137):
            seqs = [rseq(rng, W) for _ in range(2)]
            cons = [canonical_constraint(rng, s, 4) for s in seqs]
            emul.set_kernel_mode(1)
            r1 = emul.fold_constrained(seqs, cons, mfe=False)
            emul.set_kernel_mode(0)
            r = emul.fold_constrained(seqs, cons, mfe=False)
            assert r["centroid"] == r1["centroid"] and np.abs(np.asarray(r["dG"]) - np.asarray(r1["dG"])).max() < 1e-9
            for k in range(2):
                oracle.set_constraint(cons[k], None)
                o = oracle.pf(seqs[k])
                assert o["centroid"] == r["centroid"][k]
                assert abs(o["dG"] - r["dG"][k]) < 1e-9 and abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < 1e-9
    finally:
        emul.set_kernel_mode(0)
        oracle.set_constraint(None, None)


def test_rna_facade_constraints_and_shape(emul, oracle, monkeypatch):
    from scanfold_amd import RNA
    emul.load_params(params.default_params())
    monkeypatch.setattr(_lib, "_engine", emul)
    rng = np.random.default_rng(8)
    s = rseq(rng, 40)
    cons = "....((((......)))).........xxxx<<<...>>>"
    try:
        fc = RNA.fold_compound(s, RNA.md())
        fc.hc_add_from_db(cons)
        oracle.set_constraint(cons, None)
        db, e = oracle.mfe(s)
        assert fc.mfe() == (db, float(np.float32(e) / np.float32(100)))
        fc.pf()
        o = oracle.pf(s)
        assert fc.centroid()[0] == o["centroid"] and abs(fc.mean_bp_distance() - o["mean_bp_dist"]) < 1e-9
        with pytest.raises(ValueError):
            fc.hc_add_from_db("...")
        # SHAPE, Deigan: the reference's call order (pf first, then the soft constraint, then mfe: ScanFold.py:525-541)
        react = [float(x) for x in np.round(rng.uniform(0, 2.5, 40), 3)]
        react[7] = -999.0
        fc = RNA.fold_compound(s, RNA.md())
        fc.pf()
        ed_plain = fc.mean_bp_distance()
        fc.sc_add_SHAPE_deigan(react, 1.8, -0.6)
        pe = RNA.deigan_pseudo_energies(react, 1.8, -0.6, 40)
        # position k (0-based) takes list element k+1 (ViennaRNA reads the vector 1-based); the last one is missing data
        assert pe[6] == 0 and pe[39] == 0 and pe[0] == int(round(float(np.float32((1.8 * math.log(react[1] + 1) - 0.6) * 100))))
        oracle.set_constraint(None, pe)
        db, e = oracle.mfe(s)
        assert fc.mfe() == (db, float(np.float32(e) / np.float32(100)))
        assert fc.mean_bp_distance() == ed_plain
        with pytest.raises(TypeError):
            fc.sc_add_SHAPE_zarringhalam(react)  # upstream's one-argument call fails the same way
    finally:
        oracle.set_constraint(None, None)


def test_cli_constraints_follow_scanfold_scan_semantics(emul, oracle, tmp_path, monkeypatch):
    """-c: the native window's MFE / structure / centroid / ED are constrained, the z-score is not;
    windows that cut a bracket pair: error like ViennaRNA, or `--constraint-unbalanced ignore`."""
    emul.load_params(params.default_params())
    monkeypatch.setattr(_lib, "_engine", emul)
    rng = np.random.default_rng(12)
    seq = rseq(rng, 70)
    cons = "." * 10 + "((((......))))" + "x" * 6 + "<<..>>" + "." * 34
    fa = tmp_path / "t.fa"
    fa.write_text(">rec\n" + seq + "\n")
    cf = tmp_path / "cons.txt"
    cf.write_text(">rec\n" + seq + "\n" + cons + "\n")
    out = tmp_path / "o.txt"
    args = ["-i", str(fa), "-w", "30", "-s", "5", "-r", "3", "-type", "di", "--seed", "1", "-o", str(out)]
    with pytest.raises(_lib.ScanFoldHipError, match="unbalanced"):
        scanmod.main(args + ["-c", str(cf)])
    assert scanmod.main(args + ["-c", str(cf), "--constraint-unbalanced", "ignore"]) == 0
    got = out.read_text().split("\n")[1:-1]
    assert scanmod.main(args) == 0
    plain = out.read_text().split("\n")[1:-1]
    assert len(got) == len(plain) == 9
    try:
        for k, (g, p) in enumerate(zip(got, plain)):
            g, p = g.split("\t"), p.split("\t")
            assert g[:3] == p[:3] and g[4:6] == p[4:6] and g[7] == p[7]  # coordinates, z-score, p-score, sequence
            i = 5 * k
            wc = scanmod._drop_unmatched_brackets(np.frombuffer(cons[i:i + 30].encode(), dtype=np.uint8)[None, :].copy())
            wc = bytes(wc[0]).decode()
            oracle.set_constraint(wc, None)
            db, e = oracle.mfe(seq[i:i + 30])
            o = oracle.pf(seq[i:i + 30])
            assert g[3] == str(round(float(np.float32(e) / np.float32(100)), 2)) and g[8] == db and g[9] == o["centroid"]
            assert g[6] == str(round(o["mean_bp_dist"], 2))
    finally:
        oracle.set_constraint(None, None)
    bad = tmp_path / "short.txt"
    bad.write_text(">rec\n" + seq + "\n" + cons[:-3] + "\n")
    with pytest.raises(ValueError, match="same length"):
        scanmod.main(args + ["-c", str(bad)])


def test_temperature_rescale_end_to_end(emul, oracle):
    """-t 25 with a parameter set that has enthalpies: native fold at 25 C, shuffles and z-score at 37 C."""
    from par_util import par_text, synthetic_enthalpies
    base = params.default_params()
    p = params.parse_par_text(par_text(base.rec, synthetic_enthalpies(base.rec, 9)), source="synthetic.par")
    rng = np.random.default_rng(2)
    seq = rseq(rng, 64)
    try:
        emul.load_params(p)
        rows25 = scanmod.scan_record(seq, 30, 17, 3, "mono", 25, emul, seed=3)
        rows37 = scanmod.scan_record(seq, 30, 17, 3, "mono", 37, emul, seed=3)
        assert emul.params.temperature in (25.0, 37.0)
        oracle.set_params(p.at_temperature(25))
        for k, (a, b) in enumerate(zip(rows25, rows37)):
            a, b = a.rstrip("\n").split("\t"), b.rstrip("\n").split("\t")
            assert a[2] == "25" and b[2] == "37" and a[4:6] == b[4:6]  # same z / p: the shuffle background ignores -t
            s = seq[17 * k:17 * k + 30]
            db, e = oracle.mfe(s)
            o = oracle.pf(s)
            assert a[3] == str(round(float(np.float32(e) / np.float32(100)), 2)) and a[8] == db and a[9] == o["centroid"]
            assert a[6] == str(round(o["mean_bp_dist"], 2))
        # energies(seq_list, T) of the ScanFoldFunctions flavour honours the temperature (ScanFoldFunctions.py:774-789)
        emul.set_temperature(25)
        e25 = emul.mfe_batch([seq[:30], seq[30:60]])
        assert list(e25) == [oracle.mfe(seq[:30])[1], oracle.mfe(seq[30:60])[1]]
        # Boltzmann weights at 25 C come from the un-truncated rescaled doubles (ViennaRNA get_boltzmann_factors [EXT]),
        # the MFE tables from the truncated integers.  -kT ln Z is monotone in every table entry, so the exact ensemble
        # energy lies between those of the all-floor and the all-ceil integer tables; the truncated tables alone
        # (sf_params_load, no 37 C / enthalpy records) give a slightly different number.
        p25 = p.at_temperature(25)
        wins = [seq[:30], seq[30:60], seq[17:47]]
        exact = emul.pf_batch(wins)["dG"]
        assert np.allclose(exact, [oracle.pf(w)["dG"] for w in wins], rtol=0, atol=1e-8)
        tempf = (25 + 273.15) / 310.15
        bounds = {}
        for name, fn in (("floor", np.floor), ("ceil", np.ceil)):
            rec = p25.rec.copy()
            for f in params._RESCALED:
                g37, dh = p.rec37[f].astype(np.float64), p.dH[f].astype(np.float64)
                rec[f] = np.where(np.abs(p.rec37[f]) >= params.INF, p.rec37[f], fn(dh - (dh - g37) * tempf)).astype(np.int64)
            emul.load_params(params.ParamSet(rec, "bound-" + name))
            bounds[name] = emul.pf_batch(wins)["dG"]
        assert (bounds["floor"] <= exact + 1e-9).all() and (exact <= bounds["ceil"] + 1e-9).all()
        assert (bounds["ceil"] - bounds["floor"] < 0.5).all()
        emul.load_params(params.ParamSet(p25.rec.copy(), "truncated-only"))
        trunc = emul.pf_batch(wins)["dG"]
        assert 0 < np.abs(trunc - exact).max() < 0.2
        emul.load_params(base)
        with pytest.raises(NotImplementedError):
            emul.set_temperature(25)  # the reconstructed set has no enthalpies
    finally:
        oracle.set_params(base)
        emul.load_params(base)

"""T != 37 against a model that shares no code with the oracle, the CPU twin or the HIP library (round-3 advice, item 5).

tests/py_model.py restates the nearest-neighbour model in pure Python in ENERGY space: it rescales the 37 C and enthalpy
records itself, adds a structure's loops up as real numbers and sums exp(-G/kT) over every structure of a short sequence.  The
oracle (dynamic programming over Boltzmann WEIGHTS from exact_energy(), oracle/sf_oracle.c:82) must give the same ensemble
free energy and mean base-pair distance at 25, 37 and 50 C, and the same minimum free energy from the truncated tables — on
sequences chosen so that every loop class carries weight: tri- / tetra- / hexaloop hits and misses, stacks, bulges of 1 and 2,
1x1 / 2x1 / 1x2 / 2x2 / 1xn / 2x3 / generic interior loops, multiloops, exterior stems with both, one and no neighbour, GU and AU
closures.  No ViennaRNA here (absent): what this pins is that the three C/HIP restatements of the T != 37 rescale agree with
the published formula as restated a second time, not the parameter values."""
import numpy as np
import pytest

from scanfold_amd import params
import py_model
from par_util import par_text, synthetic_enthalpies

SEQS = [
    "GGGAAACCC",            # triloop-sized hairpin (size 3) on a GC helix
    "GGCGAAAGCC",           # GAAA tetraloop candidates
    "GGGGAAACUCC",          # GU inside a helix
    "GACUUCGGUC",           # UUCG
    "GGACUUUUGUCC",         # plain 4-loop, AU / GU closures
    "GGAGAAAACUCC",         # bulge / 1x1 alternatives
    "GCGAACGAAAGCGC",       # asymmetric interior loops
    "GGCAAGCAAAGCAAGCC",    # 2x2 / 2x3 neighbourhood
    "GGGAAACCCAGGGAAACCC",  # two exterior stems
    "GGAGCAAAGCAGCAAAGCACC",  # multiloop with two inner stems
    "AGGGAAACCCU",          # exterior stem with both neighbours
    "GGACGUAAGUACGUCC",     # hexaloop-sized hairpin
    "GAGUAAAUUACUC",        # AU-rich
]


@pytest.fixture(scope="module")
def pset():
    base = params.default_params()
    return params.parse_par_text(par_text(base.rec, synthetic_enthalpies(base.rec, 5)), source="synthetic.par")


@pytest.mark.parametrize("T", [25.0, 37.0, 50.0])
def test_ensemble_and_mfe_against_the_independent_python_model(oracle, pset, T):
    try:
        oracle.set_params(pset.at_temperature(T))
        n_struct, seen = 0, set()
        for seq in SEQS:
            dg, dist, count, mfe, db, classes = py_model.ensemble(pset, seq, T)
            seen |= classes
            o = oracle.pf(seq)
            assert abs(o["dG"] - dg) < 1e-9 * max(1.0, abs(dg)), (seq, T, o["dG"], dg)
            assert abs(o["mean_bp_dist"] - dist) < 1e-9, (seq, T, o["mean_bp_dist"], dist)
            odb, oe = oracle.mfe(seq)
            assert oe == mfe, (seq, T, oe, mfe, odb, db)
            # the oracle's structure evaluates to the same energy in the Python model (ties may pick another structure)
            pt = [0] * (len(seq) + 2)
            stack = []
            for k, ch in enumerate(odb, 1):
                if ch == "(":
                    stack.append(k)
                elif ch == ")":
                    a = stack.pop()
                    pt[a], pt[k] = k, a
            assert int(round(py_model.Model(pset, T, "mfe").energy(seq, pt))) == oe
            n_struct += count
        assert n_struct > 2000  # the enumeration really covers ensembles, not single structures
        want = {"hairpin", "hairpin 3", "special hairpin 4", "stack", "bulge 1", "bulge 2", "1x1", "1x2", "1x3", "1x4", "2x2", "2x3",
                "2x4", "3x3", "multiloop"}
        assert want <= seen, sorted(want - seen)
    finally:
        oracle.set_params(params.default_params())


def test_rescale_is_not_a_no_op_and_is_exact_where_it_can_be_checked_by_hand(pset):
    """One entry by hand: stack[CG][CG] at 25 C from its 37 C value and enthalpy; and the ensemble energies at the three
    temperatures differ (the test above would also pass for a model that ignored T on both sides)."""
    g, h = float(pset.rec37["stack"][1][1]), float(pset.dH["stack"][1][1])
    m = py_model.Model(pset, 25.0, "pf")
    assert abs(float(m.t["stack"][1][1]) - (h - (h - g) * 298.15 / 310.15)) < 1e-12
    assert float(py_model.Model(pset, 25.0, "mfe").t["stack"][1][1]) == float(int(h - (h - g) * 298.15 / 310.15))
    assert pset.at_temperature(25).rec["stack"][1][1] == int(h - (h - g) * 298.15 / 310.15)
    dgs = [py_model.ensemble(pset, "GGCGAAAGCC", T)[0] for T in (25.0, 37.0, 50.0)]
    assert dgs[0] < dgs[1] < dgs[2] and dgs[2] - dgs[0] > 0.5


def test_smooth_matches_its_published_shape():
    assert py_model.smooth(-20.0) == 0.0 and py_model.smooth(50.0) == 50.0
    assert abs(py_model.smooth(8.660254) - 8.660254) < 1e-5          # continuous at the upper joint
    assert py_model.smooth(-12.283697) < 1e-6                          # and at the lower one
    xs = np.linspace(-12.0, 8.5, 50)
    ys = [py_model.smooth(x) for x in xs]
    assert all(b > a for a, b in zip(ys, ys[1:]))                      # monotone in between


def test_kernel_code_against_the_independent_python_model(pset):
    """The product's own T != 37 path (sf_params_load_rescaled -> ex() -> the partition-function and MFE kernels, compiled for
    the CPU emulation) against the Python model, without the oracle in between: ensemble energy, ensemble diversity, MFE."""
    import os
    from scanfold_amd import _lib
    emul = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul", "libscanfold_emul.so")
    if not os.path.exists(emul):
        pytest.skip("tests/emul not built")
    eng = _lib.Engine(0, lib_path=emul)
    try:
        eng.load_params(pset)
        for T in (25.0, 50.0):
            eng.set_temperature(T)
            for seq in SEQS:
                dg, dist, count, mfe, db, _ = py_model.ensemble(pset, seq, T)
                o = eng.pf_batch([seq])
                assert abs(float(o["dG"][0]) - dg) < 1e-9 * max(1.0, abs(dg)), (seq, T, o["dG"][0], dg)
                assert abs(float(o["mean_bp_dist"][0]) - dist) < 1e-9, (seq, T)
                assert int(eng.mfe_batch([seq])[0]) == mfe, (seq, T)
            # and the constrained folds of the same code (sf_fold_constrained: the constraint at the pair-type seam of the kernels)
            for seq, cons in CONSTRAINED:
                dg, dist, count, mfe, db, _ = py_model.ensemble(pset, seq, T, cons=cons)
                o = eng.fold_constrained([seq], [cons])
                assert abs(float(o["dG"][0]) - dg) < 1e-9 * max(1.0, abs(dg)) and abs(float(o["mean_bp_dist"][0]) - dist) < 1e-9, (seq, cons, T)
                assert int(o["mfe"][0]) == mfe, (seq, cons, T)
    finally:
        eng.load_params(params.default_params())


CONSTRAINED = [
    ("GGGAAACCCAGGGAAACCC", "xx....<............"),   # G1, G2 unpaired; C7 may only pair downstream
    ("GGGAAACCCAGGGAAACCC", "(.......).........."),   # G1-C9 pair with each other only, nothing crosses them
    ("GGAGCAAAGCAGCAAAGCACC", ".(.................)."),
    ("GGAGCAAAGCAGCAAAGCACC", "..x.>..|....<......>."),
    ("GGCAAGCAAAGCAAGCC", ">......x......<.."),
    ("GACUUCGGUCAGGACUUUUGUCC", ".(......)....x........."),   # A2-U9 (a bracket pair of complementary bases)
]


@pytest.mark.parametrize("T", [37.0, 25.0])
def test_hard_constraints_against_the_independent_python_model(oracle, pset, T):
    """-c: the oracle applies a constraint per cell of its DP (pair-type seam: sf_oracle.c ptype()); the Python model filters whole
    structures (x unpaired, < / > pair down- / upstream only, a bracket pair pairs with itself only and nothing crosses it).
    Ensemble energy, ensemble diversity and MFE agree; every case really excludes structures."""
    try:
        oracle.set_params(pset.at_temperature(T))
        for seq, cons in CONSTRAINED:
            free = py_model.ensemble(pset, seq, T)
            dg, dist, count, mfe, db, _ = py_model.ensemble(pset, seq, T, cons=cons)
            assert 0 < count < free[2]
            oracle.set_constraint(cons, None)
            o = oracle.pf(seq)
            assert abs(o["dG"] - dg) < 1e-9 * max(1.0, abs(dg)) and abs(o["mean_bp_dist"] - dist) < 1e-9, (seq, cons, T)
            odb, oe = oracle.mfe(seq)
            assert oe == mfe, (seq, cons, T, odb, db)
            oracle.set_constraint(None, None)
    finally:
        oracle.set_constraint(None, None)
        oracle.set_params(params.default_params())


def test_deigan_stack_terms_against_the_independent_python_model(oracle, pset):
    """SHAPE (Deigan): per-nucleotide pseudo-energies charged to stacks (fc.sc_add_SHAPE_deigan, ScanFold.py:522-544) — MFE of
    the oracle and of the emulated kernel code == the minimum over all structures in the Python model; the terms change results."""
    import os
    from scanfold_amd import _lib
    rng = np.random.default_rng(8)
    emul = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emul", "libscanfold_emul.so")
    eng = _lib.Engine(0, lib_path=emul) if os.path.exists(emul) else None
    changed = 0
    try:
        oracle.set_params(pset.at_temperature(37.0))
        for seq in SEQS[5:]:
            sc = rng.integers(-120, 160, len(seq)).astype(np.int32)
            want, _ = py_model.mfe_with_shape(pset, seq, 37.0, [int(x) for x in sc])
            oracle.set_constraint(None, sc)
            assert oracle.mfe(seq)[1] == want, seq
            oracle.set_constraint(None, None)
            changed += want != oracle.mfe(seq)[1]
            if eng is not None:
                eng.load_params(pset)
                assert int(eng.fold_constrained([seq], None, sc.reshape(1, -1), pf=False)["mfe"][0]) == want, seq
        assert changed >= 3
    finally:
        oracle.set_constraint(None, None)
        oracle.set_params(params.default_params())
        if eng is not None:
            eng.load_params(params.default_params())

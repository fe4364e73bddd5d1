"""Parity tests proper: the HIP path, called through the C ABI, against the oracle on the same inputs.

Bars: integer energies (dcal/mol), structure and centroid strings, TSV bytes — bit-exact;
partition-function scalars (FP64 on both sides, different summation order) — |delta| < 1e-8.
Nothing here reads /root/reference (absent on the GPU box)."""
import json
import os
from collections import Counter

import numpy as np
import pytest

from scanfold_amd import _lib, params
from scanfold_amd import scan as scanmod
from conftest import random_seqs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PF_TOL = 1e-8


def ascii_rows(codes):
    return np.frombuffer(b"NACGU", dtype=np.uint8)[codes]


def synth_transcript(L, seed):
    return "".join("ACGU"[k] for k in np.random.default_rng(seed).integers(0, 4, L))


def test_engine_reports_a_gfx950_device(gpu_engine):
    assert "gfx950" in gpu_engine.device_name()


def test_committed_vectors(gpu_engine):
    items = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json")))["items"]
    for it in items:
        e, db = gpu_engine.mfe_trace_batch([it["seq"]])
        assert (int(e[0]), db[0]) == (it["mfe_dcal"], it["structure"]), it["seq"]
        assert int(gpu_engine.mfe_batch([it["seq"]])[0]) == it["mfe_dcal"]
        r = gpu_engine.pf_batch([it["seq"]])
        assert r["centroid"][0] == it["centroid"]
        assert abs(r["dG"][0] - it["ens_dG"]) < PF_TOL and abs(r["mean_bp_dist"][0] - it["mean_bp_dist"]) < PF_TOL
        assert abs(r["centroid_dist"][0] - it["centroid_dist"]) < PF_TOL


@pytest.mark.parametrize("W,n", [(8, 64), (9, 64), (30, 512), (77, 512), (120, 4096), (128, 256), (129, 256),
                                  (200, 512), (256, 64), (300, 16), (400, 4), (5, 8), (167, 128), (168, 384), (181, 384)])
def test_mfe_energy_parity_both_kernels(gpu_engine, oracle, W, n):
    arr = random_seqs(np.random.default_rng(W * 1000 + n), n, W)
    ref = oracle.mfe_batch(arr)
    gpu_engine.set_kernel_mode(0)
    fast = gpu_engine.mfe_batch(arr)
    gpu_engine.set_kernel_mode(1)
    full = gpu_engine.mfe_batch(arr[: min(n, 512)])
    gpu_engine.set_kernel_mode(0)
    assert (fast == ref).all(), int((fast != ref).sum())
    assert (full == ref[: len(full)]).all()


def test_biased_compositions_and_int16_overflow_fallback(gpu_engine, oracle):
    rng = np.random.default_rng(5)
    W = 120
    gc = np.frombuffer(b"GC", dtype=np.uint8)[rng.integers(0, 2, (64, W))]
    au = np.frombuffer(b"AU", dtype=np.uint8)[rng.integers(0, 2, (64, W))]
    helix = np.zeros((2, W), dtype=np.uint8)
    helix[0, :W // 2] = ord("G"); helix[0, W // 2:] = ord("C")
    helix[1] = np.frombuffer(("GC" * W)[:W].encode(), dtype=np.uint8)
    arr = np.concatenate([gc, au, helix])
    ref = oracle.mfe_batch(arr)
    assert ref.min() < -12000  # the helix rows leave the int16-safe range and must come back exact
    assert (gpu_engine.mfe_batch(arr) == ref).all()


def test_traceback_and_partition_function_parity(gpu_engine, oracle):
    for W, n in ((30, 64), (120, 128), (200, 16)):
        arr = random_seqs(np.random.default_rng(W), n, W)
        e, db = gpu_engine.mfe_trace_batch(arr)
        r = gpu_engine.pf_batch(arr)
        for k in range(n):
            s = bytes(arr[k]).decode()
            odb, oe = oracle.mfe(s)
            assert (db[k], int(e[k])) == (odb, oe)
            assert oracle.eval_structure(s, db[k]) == oe
            o = oracle.pf(s)
            assert o["centroid"] == r["centroid"][k]
            assert abs(o["dG"] - r["dG"][k]) < PF_TOL
            assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < PF_TOL
            assert abs(o["centroid_dist"] - r["centroid_dist"][k]) < PF_TOL


def test_randomised_parameter_tables(gpu_engine):
    from oracle import oracle as orc
    try:
        for seed in (0, 1):
            p = params.random_params(seed)
            if seed:
                p.rec["MLclosing"] = -150
            orc.set_params(p)
            gpu_engine.load_params(p)
            arr = random_seqs(np.random.default_rng(seed), 256, 90)
            assert (gpu_engine.mfe_batch(arr) == orc.mfe_batch(arr)).all()
            e, db = gpu_engine.mfe_trace_batch(arr[:32])
            assert db == [orc.mfe(bytes(r).decode())[0] for r in arr[:32]]
            r = gpu_engine.pf_batch(arr[:8])
            for k in range(8):
                assert abs(orc.pf(bytes(arr[k]).decode())["dG"] - r["dG"][k]) < PF_TOL
    finally:
        gpu_engine.load_params(params.default_params())


def test_n_lowercase_t_and_codes_inputs(gpu_engine, oracle):
    seqs = ["GGGGAAAANCCCCNNNNNNN", "ggggaaaaccccaaaaaaaa", "GGGGTTTTCCCCAAAAAAAA", "NNNNNNNNNNNNNNNNNNNN"]
    ref = [oracle.mfe(s)[1] for s in seqs]
    assert list(gpu_engine.mfe_batch(seqs)) == ref
    codes = np.array([[{"A": 1, "C": 2, "G": 3, "U": 4, "T": 4}.get(ch.upper(), 0) for ch in s] for s in seqs],
                     dtype=np.uint8)
    assert list(gpu_engine.mfe_batch(codes)) == ref


def test_device_shuffles_preserve_the_reference_invariants(gpu_engine):
    tr = synth_transcript(3000, 8)
    W, step, r = 120, 60, 20
    nwin = (len(tr) - W) // step + 1
    for kind in (_lib.SHUFFLE_MONO, _lib.SHUFFLE_DI):
        rows = ascii_rows(gpu_engine.shuffle_windows(tr, W, step, 0, nwin, r, kind, 2024))
        distinct = set()
        for w in range(nwin):
            nat = bytes(rows[w * (r + 1)]).decode()
            assert nat == tr[w * step:w * step + W]
            for k in range(1, r + 1):
                s = bytes(rows[w * (r + 1) + k]).decode()
                distinct.add(s)
                if kind == _lib.SHUFFLE_MONO:
                    assert Counter(s) == Counter(nat)
                else:
                    assert s[0] == nat[0] and s[-1] == nat[-1]
                    assert Counter(zip(s, s[1:])) == Counter(zip(nat, nat[1:]))
        assert len(distinct) == nwin * r  # no repeated shuffle
        part = gpu_engine.shuffle_windows(tr, W, step, 7, 5, r, kind, 2024)
        assert (ascii_rows(part) == rows[7 * (r + 1):12 * (r + 1)]).all()
        other = gpu_engine.shuffle_windows(tr, W, step, 0, 2, r, kind, 2025)
        assert (ascii_rows(other)[1] != rows[1]).any()  # the seed matters


def test_device_shuffle_equals_the_shuffle_oracle_bit_for_bit(gpu_engine, oracle):
    """sf_shuffle_kernel vs oracle/sf_shuffle_oracle.c (the reference's algorithm, ScanFold-Scan.py:87-209,248-250,
    on the product's Philox stream): cfg1, cfg2 and 10 000 random windows that contain N."""
    cases = [(synth_transcript(1000, 1), 120, 40, 0, 23, 10, 3), (synth_transcript(10000, 2), 120, 10, 0, 989, 30, 11)]
    rng = np.random.default_rng(44)
    trn = "".join("ACGUNt"[k] for k in rng.choice(6, 10119, p=[0.24, 0.24, 0.24, 0.24, 0.03, 0.01]))
    cases += [(trn, 120, 1, 0, 10000, 3, 2 ** 41 + 7), (trn, 200, 37, 5, 200, 7, 1), (trn, 16, 3, 0, 700, 2, 6)]
    for kind in (_lib.SHUFFLE_MONO, _lib.SHUFFLE_DI):
        for (tr, W, step, wb, nw, r, seed) in cases:
            dev = gpu_engine.shuffle_windows(tr, W, step, wb, nw, r, kind, seed)
            assert (dev == oracle.shuffle_windows(tr, W, step, wb, nw, r, kind, seed)).all(), (kind, W, step)


def test_device_di_shuffle_is_uniform_over_the_enumerated_set(gpu_engine):
    """The distribution that decides every z-score: on short inputs the device dinucleotide shuffle must be uniform over
    the exhaustively enumerated set of admissible sequences, and indistinguishable from the histogram of the
    reference's own dinuclShuffle (tests/golden/reference_dinucl_hist.json, 30 000 reference draws per input)."""
    from shuffle_util import chi2_limit, chi2_two_sample, chi2_uniform, codes_to_str, di_arrangements
    items = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_dinucl_hist.json")))["items"]
    for k, it in enumerate(items):
        s = it["s"]
        allowed = di_arrangements(s)
        rows = codes_to_str(gpu_engine.shuffle_windows(s, len(s), 1, 0, 1, 50000, _lib.SHUFFLE_DI, 900 + k))
        hist = Counter(rows[1:])
        assert sorted(hist) == allowed, s
        assert chi2_uniform([hist[a] for a in allowed]) < chi2_limit(len(allowed) - 1), s
        assert chi2_two_sample([hist[a] for a in allowed], [it["hist"][a] for a in allowed]) < chi2_limit(
            len(allowed) - 1), s
    # mono: uniform over the 60 distinct permutations of a 5-mer with one repeated letter
    rows = codes_to_str(gpu_engine.shuffle_windows("ACGUA", 5, 1, 0, 1, 60000, _lib.SHUFFLE_MONO, 5))[1:]
    hist = Counter(rows)
    assert len(hist) == 60 and chi2_uniform(list(hist.values())) < chi2_limit(59)


def test_mono_shuffle_is_close_to_uniform(gpu_engine):
    tr = "ACGU" * 5
    rows = gpu_engine.shuffle_windows(tr, 20, 1, 0, 1, 20000, _lib.SHUFFLE_MONO, 1)[1:]
    freq = (rows == 1).mean(axis=0)  # P(A at each position) should be 1/4 everywhere
    assert np.abs(freq - 0.25).max() < 0.02


def build_expected_rows(oracle, eng, seq, W, step, r, kind, seed):
    starts = scanmod.window_starts(len(seq), W, step)
    rows = ascii_rows(oracle.shuffle_windows(seq, W, step, 0, len(starts), r, kind, seed))  # oracle shuffles too
    E = oracle.mfe_batch(rows).reshape(len(starts), r + 1)
    structs, cens, eds = [], [], []
    for i in starts:
        frag = scanmod.transcribe(seq[i:i + W])
        structs.append(oracle.mfe(frag)[0])
        o = oracle.pf(frag)
        cens.append(o["centroid"])
        eds.append(o["mean_bp_dist"])
    return scanmod.rows_from_results(seq, starts, W, r, 37, E, structs, cens, np.array(eds))


def test_config1_tsv_bytes_equal_oracle_built_tsv(gpu_engine, oracle):
    # BASELINE config 1: 1 kb, W=120, step=40, 10 shuffles -> 23 windows / 253 folds
    seq = synth_transcript(1000, 1)
    for kind_name, kind in (("di", _lib.SHUFFLE_DI), ("mono", _lib.SHUFFLE_MONO)):
        got = scanmod.scan_record(seq, 120, 40, 10, kind_name, 37, gpu_engine, seed=3)
        exp = build_expected_rows(oracle, gpu_engine, seq, 120, 40, 10, kind, 3)
        assert len(got) == 23 and got == exp


def test_config2_all_energies_equal_oracle(gpu_engine, oracle):
    # BASELINE config 2: 10 kb, W=120, step=10, 30 shuffles -> 989 windows / 30 659 folds
    seq = synth_transcript(10000, 2)
    res = gpu_engine.scan(seq, 120, 10, 0, 989, 30, _lib.SHUFFLE_DI, 11)
    rows = ascii_rows(gpu_engine.shuffle_windows(seq, 120, 10, 0, 989, 30, _lib.SHUFFLE_DI, 11))
    assert (res["energies"].reshape(-1) == oracle.mfe_batch(rows)).all()
    for w in range(0, 989, 97):
        s = seq[w * 10:w * 10 + 120]
        assert oracle.mfe(s)[0] == res["structure"][w]
        o = oracle.pf(s)
        assert o["centroid"] == res["centroid"][w] and abs(o["mean_bp_dist"] - res["ens_div"][w]) < PF_TOL


def test_config3_size_independent_properties(gpu_engine, oracle):
    # BASELINE config 3 shape (30 kb, W=120, step=1, 100 shuffles) on a 3 000-window slice of it, plus
    # properties that do not need the oracle at full size
    seq = synth_transcript(30000, 3)
    W, r = 120, 100
    nwin_total = len(seq) - W + 1
    assert nwin_total == 29881
    lo, n = 12000, 3000
    flags = _lib.SCAN_NO_PF | _lib.SCAN_NO_TRACE
    full = gpu_engine.scan(seq, W, 1, lo, n, r, _lib.SHUFFLE_DI, 17, flags)["energies"]
    # (a) sharding / batching invariance: any sub-range reproduces the same numbers
    a = gpu_engine.scan(seq, W, 1, lo, 1000, r, _lib.SHUFFLE_DI, 17, flags)["energies"]
    b = gpu_engine.scan(seq, W, 1, lo + 1000, 2000, r, _lib.SHUFFLE_DI, 17, flags)["energies"]
    assert (np.concatenate([a, b]) == full).all()
    assert int(a.sum(dtype=np.int64) + b.sum(dtype=np.int64)) == int(full.sum(dtype=np.int64))
    # (b) native column == independent energies() call on the window strings
    nat = gpu_engine.mfe_batch([seq[i:i + W] for i in range(lo, lo + n)])
    assert (nat == full[:, 0]).all()
    # (c) oracle on a random sample of (window, shuffle) cells
    rows = ascii_rows(gpu_engine.shuffle_windows(seq, W, 1, lo + 500, 40, r, _lib.SHUFFLE_DI, 17))
    assert (oracle.mfe_batch(rows).reshape(40, r + 1) == full[500:540]).all()
    # (d) both kernels agree on everything
    gpu_engine.set_kernel_mode(1)
    try:
        slow = gpu_engine.scan(seq, W, 1, lo, 200, r, _lib.SHUFFLE_DI, 17, flags)["energies"]
    finally:
        gpu_engine.set_kernel_mode(0)
    assert (slow == full[:200]).all()
    # (e) idempotence
    again = gpu_engine.scan(seq, W, 1, lo, n, r, _lib.SHUFFLE_DI, 17, flags)["energies"]
    assert (again == full).all()


def test_bench_instantiation_cfg3_w120_step1_r100_all_outputs(gpu_engine, oracle):
    """The exact code path bench.py times (BASELINE config 3: W=120, step=1, r=100, di, flags=0: in-kernel traceback
    of every native row + partition function with inside tables shared between consecutive windows), on three
    ranges of the 30 kb transcript — its first windows, its last windows, a middle stretch; 2 650 windows, dozens
    of shared-inside runs and their boundaries.  EVERY window is compared with the oracle (structure, centroid,
    ensemble diversity, ensemble dG, native energy) and with the stand-alone kernels; all 101 energies of 96
    windows are compared with the oracle on the oracle's own shuffles.  ScanFold-Scan.py:382-389,419-423."""
    seq = synth_transcript(30000, 3)
    W, r, seed = 120, 100, 2026
    nwin_total = len(seq) - W + 1
    for lo, n in ((0, 900), (14000, 850), (nwin_total - 900, 900)):
        res = gpu_engine.scan(seq, W, 1, lo, n, r, _lib.SHUFFLE_DI, seed, 0)
        wins = [seq[i:i + W] for i in range(lo, lo + n)]
        # stand-alone kernels on the same windows
        e1, db1 = gpu_engine.mfe_trace_batch(wins)
        alone = gpu_engine.pf_batch(wins)
        assert (res["energies"][:, 0] == e1).all() and res["structure"] == db1
        assert res["centroid"] == alone["centroid"]
        assert np.allclose(res["ens_div"], alone["mean_bp_dist"], rtol=1e-12, atol=1e-12)
        assert np.allclose(res["ens_dG"], alone["dG"], rtol=1e-12, atol=1e-12)
        # the oracle on every window (one OpenMP thread per window)
        nat = np.frombuffer("".join(wins).encode(), dtype=np.uint8).reshape(n, W)
        ref = oracle.scan_windows(nat, n, 0)
        assert (ref["energies"][:, 0] == res["energies"][:, 0]).all()
        assert ref["structure"] == res["structure"]
        assert ref["centroid"] == res["centroid"]
        assert np.abs(ref["ens_div"] - res["ens_div"]).max() < PF_TOL
        for w in range(0, n, 50):
            assert abs(oracle.pf(wins[w])["dG"] - res["ens_dG"][w]) < PF_TOL
        # all r+1 energies of 32 windows per range, shuffles regenerated by the shuffle oracle
        for w0 in (0, n // 2, n - 16):
            rows = ascii_rows(oracle.shuffle_windows(seq, W, 1, lo + w0, 16, r, _lib.SHUFFLE_DI, seed))
            assert (oracle.mfe_batch(rows).reshape(16, r + 1) == res["energies"][w0:w0 + 16]).all()
    # the z-score / TSV tail on the first range, against rows built from oracle values only
    got = scanmod.scan_record(seq[:1019], W, 1, r, "di", 37, gpu_engine, seed=seed)
    assert got == build_expected_rows(oracle, gpu_engine, seq[:1019], W, 1, r, _lib.SHUFFLE_DI, seed)


def test_config5_shape_w200_r1000_slice(gpu_engine, oracle):
    # BASELINE config 5 shape (W=200, 1000 shuffles, partition function) on a 64-window slice of the 30 kb transcript:
    # 64 064 folds of 200 nt, every energy, structure, centroid and ensemble diversity.  Reference values: the oracle's
    # faster twin (oracle/sf_cpu_twin.c, itself checked against the checker in tests/test_oracle.py) for all 64 windows,
    # the checker itself (oracle/sf_oracle.c) for the first six.
    seq = synth_transcript(30000, 3)
    W, r, lo, n = 200, 1000, 777, 64
    res = gpu_engine.scan(seq, W, 1, lo, n, r, _lib.SHUFFLE_DI, 99)
    rows = ascii_rows(gpu_engine.shuffle_windows(seq, W, 1, lo, n, r, _lib.SHUFFLE_DI, 99))
    ref = oracle.twin_scan_windows(rows, n, r)
    assert (res["energies"] == ref["energies"]).all()
    assert res["structure"] == ref["structure"] and res["centroid"] == ref["centroid"]
    assert np.abs(res["ens_div"] - ref["ens_div"]).max() < PF_TOL
    chk = oracle.scan_windows(rows[:6 * (r + 1)], 6, r)
    assert (res["energies"][:6] == chk["energies"]).all() and res["structure"][:6] == chk["structure"]
    assert res["centroid"][:6] == chk["centroid"] and np.abs(res["ens_div"][:6] - chk["ens_div"]).max() < PF_TOL


def test_edge_shapes(gpu_engine, oracle):
    # L == W (one window), step larger than the remainder, r = 0 and r = 1, the smallest LDS-kernel width
    seq = synth_transcript(120, 21)
    res = gpu_engine.scan(seq, 120, 1, 0, 1, 0, _lib.SHUFFLE_DI, 1)
    assert res["energies"].shape == (1, 1) and int(res["energies"][0, 0]) == oracle.mfe(seq)[1]
    assert res["structure"][0] == oracle.mfe(seq)[0]
    rows = scanmod.scan_record(seq + "ACGU" * 3, 120, 50, 1, "mono", 37, gpu_engine, seed=2)
    assert len(rows) == 1 and rows[0].split("\t")[:2] == ["1", "120"]
    seq16 = synth_transcript(400, 22)
    res = gpu_engine.scan(seq16, 16, 3, 0, 129, 4, _lib.SHUFFLE_MONO, 3)
    rows16 = ascii_rows(gpu_engine.shuffle_windows(seq16, 16, 3, 0, 129, 4, _lib.SHUFFLE_MONO, 3))
    assert (res["energies"].reshape(-1) == oracle.mfe_batch(rows16)).all()
    # a window containing N: device di-shuffle treats N as a fifth symbol, folds treat it as non-pairing
    seqn = seq[:50] + "NNNN" + seq[54:]
    res = gpu_engine.scan(seqn, 120, 1, 0, 1, 8, _lib.SHUFFLE_DI, 4)
    rowsn = ascii_rows(gpu_engine.shuffle_windows(seqn, 120, 1, 0, 1, 8, _lib.SHUFFLE_DI, 4))
    assert (res["energies"].reshape(-1) == oracle.mfe_batch(rowsn)).all()
    for k in range(1, 9):
        s = bytes(rowsn[k]).decode()
        assert Counter(zip(s, s[1:])) == Counter(zip(seqn, seqn[1:])) and s.count("N") == 4
    # empty batches are fine
    assert len(gpu_engine.mfe_batch(np.zeros((0, 120), dtype=np.uint8))) == 0
    assert gpu_engine.scan(seq, 120, 1, 0, 0, 5, _lib.SHUFFLE_DI, 1)["energies"].shape == (0, 6)


def test_planted_hairpin_gets_a_negative_zscore(gpu_engine):
    rng = np.random.default_rng(9)
    bg = "".join("ACGU"[k] for k in rng.choice(4, 400, p=[0.3, 0.2, 0.2, 0.3]))
    stem = "GGCGCGGCACCGUCCGCGGAACAAACGG"
    comp = stem[::-1].translate(str.maketrans("ACGU", "UGCA"))
    seq = bg[:140] + stem + "GAAA" + comp + bg[200:]
    rows = scanmod.scan_record(seq, 120, 20, 50, "di", 37, gpu_engine, seed=1)
    z = [float(r.split("\t")[4]) for r in rows]
    assert min(z) < -2.0


def test_torch_device_pointer_path_equals_host_path(gpu_engine):
    import torch
    seq = synth_transcript(2000, 4)
    W, step, r = 120, 9, 12
    nwin = (len(seq) - W) // step + 1
    host = gpu_engine.scan(seq, W, step, 0, nwin, r, _lib.SHUFFLE_DI, 5)
    dev = torch.device("cuda:0")
    tr = torch.tensor(list(seq.encode()), dtype=torch.uint8, device=dev)
    en = torch.empty((nwin, r + 1), dtype=torch.int32, device=dev)
    db = torch.zeros((nwin, W + 1), dtype=torch.uint8, device=dev)
    cen = torch.zeros((nwin, W + 1), dtype=torch.uint8, device=dev)
    div = torch.zeros(nwin, dtype=torch.float64, device=dev)
    dG = torch.zeros(nwin, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    gpu_engine.scan_dev(tr.data_ptr(), len(seq), W, step, 0, nwin, r, _lib.SHUFFLE_DI, 5, 0, en.data_ptr(),
                        db.data_ptr(), cen.data_ptr(), div.data_ptr(), dG.data_ptr(), st)
    torch.cuda.synchronize()
    assert (en.cpu().numpy() == host["energies"]).all()
    assert [bytes(x[:W]).decode() for x in db.cpu().numpy()] == host["structure"]
    assert [bytes(x[:W]).decode() for x in cen.cpu().numpy()] == host["centroid"]
    assert np.array_equal(div.cpu().numpy(), host["ens_div"])


def test_scanfoldfunctions_surface_on_gpu(gpu_engine, oracle):
    from scanfold_amd import RNA, functions as sff, scan_functions as sfn
    import random
    random.seed(3)
    frag = synth_transcript(120, 6)
    seqlist = [frag] + sff.scramble(frag, 10, "di")
    el = sff.energies(seqlist, 37, "rnafold")
    assert el == [float(np.float32(oracle.mfe(s)[1]) / np.float32(100)) for s in seqlist]
    assert sff.rna_folder((frag, 37, "rnafold")) == el[0] == sfn.rna_folder(frag) == sfn.energies([frag])[0]
    assert sff.multiprocessing(sff.rna_folder, [(s, 37, "rnafold") for s in seqlist], 12) == el
    with pytest.raises(UnboundLocalError):
        sff.energies(seqlist, 37, "rnastructure")
    with pytest.raises(NotImplementedError):
        sff.energies(seqlist, 25, "rnafold")
    assert isinstance(sff.zscore_function(el, 10), float) and 0.0 <= sff.pvalue_function(el, 10) <= 1.0
    fc = RNA.fold_compound(frag, RNA.md())
    structure, mfe = fc.mfe()
    assert (structure, mfe) == (oracle.mfe(frag)[0], el[0]) == RNA.fold(frag)
    fc.pf()
    o = oracle.pf(frag)
    assert fc.centroid()[0] == o["centroid"] and abs(fc.mean_bp_distance() - o["mean_bp_dist"]) < PF_TOL


def test_cli_writes_the_reference_named_file(gpu_engine, tmp_path):
    fa = tmp_path / "t.fa"
    seq = synth_transcript(400, 12)
    fa.write_text(">r1 test\n" + seq[:200] + "\n" + seq[200:] + "\n>short\nACGU\n")
    assert scanmod.main(["-i", str(fa), "-w", "120", "-s", "40", "-r", "10", "-type", "di", "--seed", "4"]) == 0
    out = tmp_path / "t.fa.forward.win_120.stp_40.rnd_10.shfl_di.txt"
    lines = out.read_text().split("\n")
    assert lines[0] == scanmod.header_line("r1").rstrip("\n")
    assert len(lines) == 1 + 8 + 1 and lines[1].split("\t")[:3] == ["1", "120", "37"]
    f = lines[1].split("\t")
    assert len(f) == 10 and len(f[7]) == len(f[8]) == len(f[9]) == 120


@pytest.mark.gpu
def test_max_bp_span_matches_oracle(gpu_engine, oracle):
    """sf_set_max_bp_span (RNA.md().max_bp_span, ScanFold.py:214-215): every MFE kernel, traceback and PF."""
    rng = np.random.default_rng(99)
    try:
        for W, span, n in ((120, 40, 600), (100, 100, 300), (200, 70, 120), (300, 90, 8)):
            arr = random_seqs(rng, n, W)
            oracle.set_max_bp_span(span)
            gpu_engine.set_max_bp_span(span)
            ref = oracle.mfe_batch(arr)
            for mode in (0, 1):
                gpu_engine.set_kernel_mode(mode)
                got = gpu_engine.mfe_batch(arr[: (64 if mode == 1 else n)])
                assert (got == ref[: len(got)]).all(), (W, span, mode)
            gpu_engine.set_kernel_mode(0)
            m = min(n, 40)
            e, db = gpu_engine.mfe_trace_batch(arr[:m])
            r = gpu_engine.pf_batch(arr[:m])
            for k in range(m):
                s = bytes(arr[k]).decode()
                assert (db[k], e[k]) == oracle.mfe(s)
                o = oracle.pf(s)
                assert o["centroid"] == r["centroid"][k] and abs(o["dG"] - r["dG"][k]) < PF_TOL
                assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < PF_TOL
        # a span >= W changes nothing
        arr = random_seqs(rng, 256, 120)
        oracle.set_max_bp_span(0)
        gpu_engine.set_max_bp_span(120)
        assert (gpu_engine.mfe_batch(arr) == oracle.mfe_batch(arr)).all()
    finally:
        gpu_engine.set_kernel_mode(0)
        gpu_engine.set_max_bp_span(0)
        oracle.set_max_bp_span(0)


def test_scan_step_one_shares_inside_tables(gpu_engine, oracle):
    """sf_scan with step 1 lets consecutive native windows share their partition-function inside tables (runs of up
    to 16 windows per workgroup once there are more windows than CUs).  Every window must still equal its
    stand-alone fold: against the oracle, and against the same kernel run on the windows as independent rows."""
    W = 60
    tr = synth_transcript(W + 1400 - 1, 77)
    tr = tr[:500] + "N" + tr[501:]
    nwin = len(tr) - W + 1
    res = gpu_engine.scan(tr, W, 1, 0, nwin, 1, 1, 5)
    wins = [tr[w:w + W] for w in range(nwin)]
    alone = gpu_engine.pf_batch(wins)
    assert res["centroid"] == alone["centroid"]
    assert np.allclose(res["ens_div"], alone["mean_bp_dist"], rtol=1e-12, atol=1e-12)
    assert np.allclose(res["ens_dG"], alone["dG"], rtol=1e-12, atol=1e-12)
    for w in range(0, nwin, 7):
        o = oracle.pf(wins[w])
        assert o["centroid"] == res["centroid"][w], w
        assert abs(o["mean_bp_dist"] - res["ens_div"][w]) < PF_TOL and abs(o["dG"] - res["ens_dG"][w]) < PF_TOL, w


def test_constrained_native_folds_match_oracle(gpu_engine, oracle):
    """sf_fold_constrained (fc.hc_add_from_db / fc.sc_add_SHAPE_deigan, ScanFold-Scan.py:405-418; ScanFold.py:508-544):
    a per-window constraint at the pair-type seam of the LDS kernels (kernel mode 0: sf_mfe_fast_kernel / sf_pf_lds_kernel, HC
    instantiations — batches whose bracket pairs can all pair) and of the general int32 / FP64 kernels (mode 1, and batches
    with a bracket pair of non-complementary bases); 300 windows of W=120 and smaller shapes, both modes equal, both == oracle."""
    from test_constraints import canonical_constraint, random_constraint, rseq
    rng = np.random.default_rng(77)
    try:
        for W, n, use_sc, canon in ((120, 300, False, True), (120, 300, False, False), (120, 60, True, True), (45, 64, True, True),
                                    (100, 40, True, False), (200, 12, False, True), (160, 12, True, True), (250, 6, False, True),
                                    (121, 12, False, True)):  # 120 < W <= 250: sf_pf_fast_kernel's HC instantiation
            seqs = [rseq(rng, W) for _ in range(n)]
            cons = [canonical_constraint(rng, s, 4) for s in seqs] if canon else [random_constraint(rng, W, 4) for _ in range(n)]
            sc = rng.integers(-60, 40, (n, W)).astype(np.int32) if use_sc else None
            gpu_engine.set_kernel_mode(1)
            r1 = gpu_engine.fold_constrained(seqs, cons, sc)
            gpu_engine.set_kernel_mode(0)
            r = gpu_engine.fold_constrained(seqs, cons, sc)
            assert r["structure"] == r1["structure"] and (r["mfe"] == r1["mfe"]).all() and r["centroid"] == r1["centroid"]
            assert np.abs(np.asarray(r["dG"]) - np.asarray(r1["dG"])).max() < PF_TOL
            assert np.abs(np.asarray(r["mean_bp_dist"]) - np.asarray(r1["mean_bp_dist"])).max() < PF_TOL
            for k in range(0, n, 1 if n <= 64 else 5):
                oracle.set_constraint(cons[k], None if sc is None else sc[k])
                assert oracle.mfe(seqs[k]) == (r["structure"][k], int(r["mfe"][k])), (W, k)
                oracle.set_constraint(cons[k], None)
                o = oracle.pf(seqs[k])
                assert o["centroid"] == r["centroid"][k]
                assert abs(o["dG"] - r["dG"][k]) < PF_TOL and abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < PF_TOL
        oracle.set_constraint(None, None)
        seqs = [rseq(rng, 120) for _ in range(50)]
        e, db = gpu_engine.mfe_trace_batch(seqs)
        r = gpu_engine.fold_constrained(seqs, ["." * 120] * 50)
        assert (r["mfe"] == e).all() and r["structure"] == db
        with pytest.raises(_lib.ScanFoldHipError, match="unbalanced"):
            gpu_engine.fold_constrained(seqs[:2], ["." * 120, "(" + "." * 119])
        # the CLI path: constrained native fold, unconstrained z-score
        seq = synth_transcript(400, 5)
        cons = ("." * 30 + "((((((......))))))" + "xxxx" + "<<<<....>>>>" + "." * 336)[:400]
        rows = scanmod.scan_record(seq, 120, 20, 10, "di", 37, gpu_engine, seed=2, constraints=cons, unbalanced="ignore")
        plain = scanmod.scan_record(seq, 120, 20, 10, "di", 37, gpu_engine, seed=2)
        for k, (a, b) in enumerate(zip(rows, plain)):
            a, b = a.rstrip("\n").split("\t"), b.rstrip("\n").split("\t")
            assert a[4:6] == b[4:6]
            wc = bytes(scanmod._drop_unmatched_brackets(
                np.frombuffer(cons[20 * k:20 * k + 120].encode(), dtype=np.uint8)[None, :].copy())[0]).decode()
            oracle.set_constraint(wc, None)
            db1, e1 = oracle.mfe(seq[20 * k:20 * k + 120])
            assert a[8] == db1 and a[3] == str(round(float(np.float32(e1) / np.float32(100)), 2))
    finally:
        oracle.set_constraint(None, None)


def test_temperature_rescale_on_gpu(gpu_engine, oracle):
    """md.temperature != 37 (ScanFold-Scan.py:70-71; ScanFoldFunctions.py:776-777) with a parameter file that has
    enthalpy sections (synthetic, in the published layout): every kernel of the hot path at 25 C and 50 C."""
    from par_util import par_text, synthetic_enthalpies
    base = params.default_params()
    p = params.parse_par_text(par_text(base.rec, synthetic_enthalpies(base.rec, 9)), source="synthetic.par")
    rng = np.random.default_rng(25)
    try:
        gpu_engine.load_params(p)
        for T in (25, 50):
            gpu_engine.set_temperature(T)
            oracle.set_params(p.at_temperature(T))
            arr = random_seqs(rng, 1500, 120)
            assert (gpu_engine.mfe_batch(arr) == oracle.mfe_batch(arr)).all(), T
            e, db = gpu_engine.mfe_trace_batch(arr[:40])
            r = gpu_engine.pf_batch(arr[:40])
            for k in range(40):
                s = bytes(arr[k]).decode()
                assert (db[k], int(e[k])) == oracle.mfe(s)
                o = oracle.pf(s)
                assert o["centroid"] == r["centroid"][k] and abs(o["dG"] - r["dG"][k]) < PF_TOL
                assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < PF_TOL
        gpu_engine.load_params(base)
        with pytest.raises(NotImplementedError):
            gpu_engine.set_temperature(25)
    finally:
        oracle.set_params(base)
        gpu_engine.load_params(base)


def test_config5_w200_step1_320_consecutive_windows(gpu_engine, oracle):
    """BASELINE config 5's native-window path — W=200, step=1: in-kernel traceback of the folded-layout MFE kernel and
    the device-table partition-function kernel — on 320 consecutive windows, every one against the oracle."""
    seq = synth_transcript(30000, 3)
    W, lo, n, r = 200, 9000, 320, 3
    res = gpu_engine.scan(seq, W, 1, lo, n, r, _lib.SHUFFLE_DI, 7)
    wins = [seq[i:i + W] for i in range(lo, lo + n)]
    ref = oracle.scan_windows(np.frombuffer("".join(wins).encode(), dtype=np.uint8).reshape(n, W), n, 0)
    assert (ref["energies"][:, 0] == res["energies"][:, 0]).all()
    assert ref["structure"] == res["structure"] and ref["centroid"] == res["centroid"]
    assert np.abs(ref["ens_div"] - res["ens_div"]).max() < PF_TOL
    rows = ascii_rows(oracle.shuffle_windows(seq, W, 1, lo, n, r, _lib.SHUFFLE_DI, 7))
    assert (oracle.mfe_batch(rows).reshape(n, r + 1) == res["energies"]).all()


def test_motif_refolds_on_gpu(gpu_engine, oracle, tmp_path, monkeypatch):
    """ScanFold.py:1726-1776 on the engine: constrained MFE / ensemble diversity of each extracted motif and the
    z-score of its unconstrained energy against host-shuffled copies, every number re-derived with the oracle."""
    import random
    from scanfold_amd import motifs, functions as sff
    monkeypatch.setattr(_lib, "_engine", gpu_engine)
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(12)
    seq, line = "", ""
    for k in range(6):
        stem = "".join("ACGU"[v] for v in rng.integers(0, 4, 5 + k)) + "GC"
        comp = stem[::-1].translate(str.maketrans("ACGU", "UGCA"))
        loop = "".join("ACGU"[v] for v in rng.integers(0, 4, 4 + 3 * k))
        gap = "".join("ACGU"[v] for v in rng.integers(0, 4, 3 + k))
        seq += gap + stem + loop + comp
        line += "." * len(gap) + "(" * len(stem) + "." * len(loop) + ")" * len(stem)
    ms = motifs.extract_structures(line + ".\n", seq + "A")
    assert len(ms) == 6
    random.seed(5)
    recs = motifs.refold_motifs("rec", ms, "di", "x.gff3")
    random.seed(5)
    try:
        for m, rec in zip(ms, recs):
            oracle.set_constraint(m.structure, None)
            db, e = oracle.mfe(m.sequence)
            assert (rec["structure"], rec["mfe"]) == (db, round(float(np.float32(e) / np.float32(100)), 2))
            assert rec["ed"] == round(oracle.pf(m.sequence)["mean_bp_dist"], 2)
            oracle.set_constraint(None, None)
            seqlist = [m.sequence] + sff.scramble(m.sequence, 100, "di")
            el = [float(np.float32(v) / np.float32(100)) for v in oracle.mfe_batch(seqlist)]
            assert rec["zscore"] == round(sff.zscore_function(el, 100), 2)
            assert rec["pvalue"] == round(sff.pvalue_function(el, 100), 2)
    finally:
        oracle.set_constraint(None, None)
    assert len(open("x.gff3").read().split("\n")) == 7


def test_device_pair_tabulation_equals_host_grouping(gpu_engine):
    """sf_tabulate_pairs (ScanFold-Fold.py:583-682,704-760): groups, first windows and numpy-order sums bit-equal to the
    numpy grouping of scanfold_amd.fold on a real scan (W=120, step 1) and on synthetic tables with groups of more than
    128 and 256 windows; the resident table of sf_scan_dev is read in place."""
    import torch
    from test_fold import _sorted_groups, _synthetic_table
    from scanfold_amd import fold, scan_functions as sf
    seq = synth_transcript(700, 21)
    W, r = 120, 12
    n = len(seq) - W + 1
    res = gpu_engine.scan(seq, W, 1, 0, n, r, 1, 5, raw=True)
    E = res["energies"].astype(np.float64) / 100.0
    z = np.round((E[:, 0] - E.mean(axis=1)) / np.maximum(E.std(axis=1), 1e-9), 2)
    starts = np.arange(1, n + 1)
    structs = [bytes(row[:W]).decode() for row in res["structure"]]
    table = fold.ScanTable("rec", starts, np.round(E[:, 0], 1), z, np.round(res["ens_div"], 2), [seq[s - 1:s - 1 + W] for s in starts],
                           structs)
    rng = np.random.default_rng(3)
    tables = [table, _synthetic_table(rng, 330, 200, 1, n_structs=3, open_prob=0.1),
              _synthetic_table(rng, 420, 400, 1, n_structs=2, open_prob=0.05), _synthetic_table(rng, 40, 33, 7, start=5)]
    for t in tables:
        host = _sorted_groups(fold.Tabulation(t).groups())
        dev = fold.DeviceTabulation(t, gpu_engine).groups()
        for a, b in zip(host, dev):
            assert np.array_equal(a, np.asarray(b))
    # the structure table where sf_scan_dev left it (rows of W + 1 bytes), never copied to the host
    dev = torch.device("cuda", 0)
    d_tr = torch.tensor(np.frombuffer(seq.encode(), dtype=np.uint8), device=dev)
    d_en = torch.empty((n, r + 1), dtype=torch.int32, device=dev)
    d_db = torch.zeros((n, W + 1), dtype=torch.uint8, device=dev)
    d_cen = torch.zeros((n, W + 1), dtype=torch.uint8, device=dev)
    d_div = torch.zeros(n, dtype=torch.float64, device=dev)
    d_dg = torch.zeros(n, dtype=torch.float64, device=dev)
    gpu_engine.scan_dev(d_tr.data_ptr(), len(seq), W, 1, 0, n, r, 1, 5, 0, d_en.data_ptr(), d_db.data_ptr(),
                        d_cen.data_ptr(), d_div.data_ptr(), d_dg.data_ptr())
    assert gpu_engine.last_status() == 0
    g = gpu_engine.tabulate_pairs(d_db.data_ptr(), table.starts, table.z, table.mfe, table.ed, W=W, row_stride=W + 1,
                                  on_device=True)
    host = _sorted_groups(fold.Tabulation(table).groups())
    for a, key in zip(host, ("k", "j", "windows", "first_window", "sum_z", "sum_mfe", "sum_ed")):
        assert np.array_equal(a, g[key])
    with pytest.raises(_lib.ScanFoldHipError, match="scan table"):
        gpu_engine.tabulate_pairs(["((..", "...."], [1, 2], np.zeros(2), np.zeros(2), np.zeros(2))


def test_dynamic_fold_distribution_large_batches(gpu_engine, oracle):
    """The batched MFE kernel hands folds to its persistent workgroups through a device-wide counter, so which
    workgroup folds what — and after which other fold — changes from run to run.  Batches large enough for the fold
    indices to exceed 16 bits (an index word once sat where the short-diagonal code reads past the rolling tables):
    two runs must agree everywhere, and a sample spread over the whole batch must match the oracle."""
    rng = np.random.default_rng(2026)
    for W, n in ((200, 70000), (120, 140000), (131, 40000), (40, 140000)):
        arr = random_seqs(rng, n, W)
        e1 = gpu_engine.mfe_batch(arr)
        e2 = gpu_engine.mfe_batch(arr)
        assert (e1 == e2).all(), (W, int((e1 != e2).sum()))
        pick = np.unique(np.concatenate([rng.integers(0, n, 900), np.arange(n - 300, n), np.arange(32768 - 150, 32768 + 150)]))
        assert (oracle.mfe_batch(arr[pick]) == e1[pick]).all(), W

"""Kernel LOGIC on the CPU: the kernel sources compiled against tests/emul/hip_emul.h vs the oracle.

This is test infrastructure (the product never loads the emulation build); it lets the no-GPU suite catch
indexing mistakes in the HIP sources before a GPU run, and it is the build the sanitizers run on.
The real parity tests are tests/test_gpu_parity.py (-m gpu)."""
from collections import Counter

import numpy as np
import pytest

from scanfold_amd import params
from conftest import random_seqs


@pytest.fixture(scope="module")
def emul():
    from emul_engine import emul_engine
    return emul_engine()


def ascii_rows(codes):
    return np.frombuffer(b"NACGU", dtype=np.uint8)[codes]


def test_fast_and_full_mfe_kernels_match_oracle(emul, oracle):
    emul.load_params(params.default_params())
    rng = np.random.default_rng(0)
    for W, n in ((8, 8), (13, 10), (37, 8), (64, 6), (120, 4), (200, 2)):
        arr = random_seqs(rng, n, W)
        ref = oracle.mfe_batch(arr)
        emul.set_kernel_mode(0)
        assert (emul.mfe_batch(arr) == ref).all(), W
        emul.set_kernel_mode(1)
        assert (emul.mfe_batch(arr) == ref).all(), W
        emul.set_kernel_mode(0)


def test_fast_kernel_odd_widths_and_traceback(emul, oracle):
    """Widths around the layout seams of the LDS kernel (folded fML rectangle, wave-group boundaries)."""
    emul.load_params(params.default_params())
    rng = np.random.default_rng(11)
    for W in (16, 61, 67, 99, 121, 127, 128, 131):
        arr = random_seqs(rng, 3, W)
        e, db = emul.mfe_trace_batch(arr)
        for k in range(len(arr)):
            assert (db[k], e[k]) == oracle.mfe(bytes(arr[k]).decode()), (W, k)


def test_fast_kernel_split_steps_on_many_sequences(emul, oracle):
    """Both helper flavours of the narrow kernel on enough sequences for rare orderings to show (the emulation runs the
    lanes of a wave one after the other, starting anywhere: a lane that publishes into its neighbour's exchange slot
    before the neighbour has read it is found here, not on the GPU): merged helper (W < 118) and one helper per
    diagonal (W >= 118)."""
    emul.load_params(params.default_params())
    rng = np.random.default_rng(77)
    for W in (80, 100, 117, 118, 120, 124, 128):
        arr = random_seqs(rng, 24, W)
        assert (emul.mfe_batch(arr) == oracle.mfe_batch(arr)).all(), W


def test_fast_kernel_four_wave_groups_with_helper_waves(emul, oracle):
    """W > 128: four waves per diagonal; from the diagonal where the cells fit the two middle waves on, the outer waves
    mirror them (special loops + a share of the multiloop split).  Widths on both sides of every seam, biased
    compositions (long helices: the multiloop split decides), MFE and traceback against the oracle."""
    emul.load_params(params.default_params())
    rng = np.random.default_rng(23)
    # (167 / 168: the narrow phase — one main + one helper wave per diagonal once a diagonal fits one wave — exists from W = 168 on;
    # 200: its instantiation also hands the idle waves 80 % of the multiloop split)
    for W in (129, 130, 146, 167, 168, 177, 200, 223, 256):
        arr = random_seqs(rng, 2, W)
        gc = np.frombuffer(b"GGGCCCAU", dtype=np.uint8)[rng.integers(0, 8, (1, W))]
        arr = np.concatenate([arr, gc])
        e, db = emul.mfe_trace_batch(arr)
        for k in range(len(arr)):
            assert (db[k], e[k]) == oracle.mfe(bytes(arr[k]).decode()), (W, k)


def test_int16_overflow_falls_back_to_exact_kernel(emul, oracle):
    W = 100
    arr = np.zeros((2, W), dtype=np.uint8)
    arr[0, :W // 2] = ord("G"); arr[0, W // 2:] = ord("C")
    arr[1] = np.frombuffer(("GC" * W)[:W].encode(), dtype=np.uint8)
    ref = oracle.mfe_batch(arr)
    assert ref.min() < -12000
    assert (emul.mfe_batch(arr) == ref).all()


def test_traceback_and_partition_function(emul, oracle):
    rng = np.random.default_rng(1)
    for W in (30, 90):
        arr = random_seqs(rng, 4, W)
        e, db = emul.mfe_trace_batch(arr)
        r = emul.pf_batch(arr)
        for k in range(len(arr)):
            s = bytes(arr[k]).decode()
            odb, oe = oracle.mfe(s)
            o = oracle.pf(s)
            assert (db[k], e[k]) == (odb, oe)
            assert abs(o["dG"] - r["dG"][k]) < 1e-9 and o["centroid"] == r["centroid"][k]
            assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < 1e-9
            assert abs(o["centroid_dist"] - r["centroid_dist"][k]) < 1e-9


def test_partition_function_at_the_lane_boundaries(emul, oracle):
    """sf_pf_lds_kernel keeps per-column state a lane per column / row in two registers (columns 1..64 and 65..128): the exterior
    walks and the blocked R1 sums change register at 64 / 65, the exterior-loop table moves between two LDS areas at W = 92.
    Widths on both sides of each boundary against the oracle (the GPU every-width sweep covers all of 16..256 on hardware)."""
    rng = np.random.default_rng(64)
    for W in (63, 64, 65, 66, 91, 92, 100):
        arr = random_seqs(rng, 2, W)
        r = emul.pf_batch(arr)
        for k in range(len(arr)):
            o = oracle.pf(bytes(arr[k]).decode())
            assert abs(o["dG"] - r["dG"][k]) < 1e-9 and o["centroid"] == r["centroid"][k], (W, k)
            assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < 1e-9 and abs(o["centroid_dist"] - r["centroid_dist"][k]) < 1e-9, (W, k)


def test_randomised_tables_catch_index_order(emul):
    from oracle import oracle as orc
    rng = np.random.default_rng(2)
    p = params.random_params(7)
    orc.set_params(p)
    emul.load_params(p)
    arr = random_seqs(rng, 6, 60)
    assert (emul.mfe_batch(arr) == orc.mfe_batch(arr)).all()
    e, db = emul.mfe_trace_batch(arr)
    assert db == [orc.mfe(bytes(r).decode())[0] for r in arr]
    emul.load_params(params.default_params())


def test_n_and_lowercase_and_t(emul, oracle):
    seqs = ["GGGGAAAANCCCCNNNNNNN", "ggggaaaaccccaaaaaaaa", "GGGGTTTTCCCCAAAAAAAA", "NNNNNNNNNNNNNNNNNNNN"]
    assert list(emul.mfe_batch(seqs)) == [oracle.mfe(s)[1] for s in seqs]


def test_device_shuffles_and_scan(emul, oracle):
    rng = np.random.default_rng(3)
    tr = "".join("ACGU"[k] for k in rng.integers(0, 4, 260))
    W, step, r = 50, 30, 6
    nwin = (len(tr) - W) // step + 1
    for kind in (0, 1):
        rows = ascii_rows(emul.shuffle_windows(tr, W, step, 0, nwin, r, kind, 99))
        for w in range(nwin):
            nat = bytes(rows[w * (r + 1)]).decode()
            assert nat == tr[w * step:w * step + W]
            for k in range(1, r + 1):
                s = bytes(rows[w * (r + 1) + k]).decode()
                if kind == 0:
                    assert Counter(s) == Counter(nat)
                else:
                    assert s[0] == nat[0] and s[-1] == nat[-1] and Counter(zip(s, s[1:])) == Counter(zip(nat, nat[1:]))
        # a window's shuffles do not depend on batching
        part = emul.shuffle_windows(tr, W, step, 2, 3, r, kind, 99)
        assert (ascii_rows(part) == rows[2 * (r + 1):5 * (r + 1)]).all()
    res = emul.scan(tr, W, step, 0, nwin, r, 1, 99)
    rows = ascii_rows(emul.shuffle_windows(tr, W, step, 0, nwin, r, 1, 99))
    assert (res["energies"].reshape(-1) == oracle.mfe_batch(rows)).all()
    for w in range(nwin):
        s = tr[w * step:w * step + W]
        assert oracle.mfe(s)[0] == res["structure"][w]
        assert oracle.pf(s)["centroid"] == res["centroid"][w]


def test_scan_step_one_shares_inside_tables(emul, oracle):
    """sf_scan with step 1: consecutive native windows reuse the inside tables of their predecessor (shifted by one
    row and column, one new column computed).  Centroid, ensemble diversity and ensemble energy must equal the
    stand-alone fold of every window — including windows with an N and the first / last window of the transcript."""
    rng = np.random.default_rng(11)
    for W, L in ((40, 75), (31, 64), (120, 131)):
        tr = "".join("ACGU"[k] for k in rng.integers(0, 4, L))
        if W == 40:
            tr = tr[:20] + "N" + tr[21:]
        nwin = L - W + 1
        res = emul.scan(tr, W, 1, 0, nwin, 1, 1, 5)
        for w in range(nwin):
            o = oracle.pf(tr[w:w + W])
            assert o["centroid"] == res["centroid"][w], (W, w)
            assert abs(o["mean_bp_dist"] - res["ens_div"][w]) < 1e-9, (W, w)
            assert abs(o["dG"] - res["ens_dG"][w]) < 1e-9, (W, w)
        # a sub-range of windows (different run boundaries) gives the same numbers
        part = emul.scan(tr, W, 1, 3, nwin - 5, 1, 1, 5)
        assert part["centroid"] == res["centroid"][3:nwin - 2]
        assert np.allclose(part["ens_div"], res["ens_div"][3:nwin - 2], rtol=0, atol=1e-9)
    # windows more than one nucleotide apart: `step` new columns per window
    tr = "".join("ACGU"[k] for k in rng.integers(0, 4, 150))
    for W, step in ((48, 2), (48, 7), (40, 10)):
        nwin = (len(tr) - W) // step + 1
        res = emul.scan(tr, W, step, 0, nwin, 1, 1, 5)
        for w in range(nwin):
            o = oracle.pf(tr[w * step:w * step + W])
            assert o["centroid"] == res["centroid"][w], (W, step, w)
            assert abs(o["mean_bp_dist"] - res["ens_div"][w]) < 1e-9 and abs(o["dG"] - res["ens_dG"][w]) < 1e-9, (W, step, w)


def test_bad_arguments_return_status(emul):
    from scanfold_amd._lib import ScanFoldHipError
    with pytest.raises(ScanFoldHipError):
        emul.mfe_batch(np.zeros((1, 500), dtype=np.uint8))  # W > SF_MAX_W
    with pytest.raises(ScanFoldHipError):
        emul.scan("ACGU" * 10, 30, 1, 0, 50, 2, 1, 0)  # windows run past the transcript
    with pytest.raises(ScanFoldHipError):
        emul.scan("ACGU" * 10, 30, 1, 0, 2, 2, 7, 0)  # unknown shuffle kind
    assert len(emul.mfe_batch(np.zeros((0, 30), dtype=np.uint8))) == 0


def test_max_bp_span_every_kernel(emul, oracle):
    """sf_set_max_bp_span: MFE (every kernel mode), traceback and partition function follow the oracle."""
    emul.load_params(params.default_params())
    rng = np.random.default_rng(5)
    try:
        for W, span in ((40, 12), (70, 30), (120, 50), (140, 45)):  # 140: the device-table PF kernel
            arr = random_seqs(rng, 3, W)
            oracle.set_max_bp_span(span)
            emul.set_max_bp_span(span)
            ref = oracle.mfe_batch(arr)
            unlimited = None
            for mode in (0, 1):
                emul.set_kernel_mode(mode)
                assert (emul.mfe_batch(arr) == ref).all(), (W, span, mode)
            emul.set_kernel_mode(0)
            e, db = emul.mfe_trace_batch(arr)
            r = emul.pf_batch(arr)
            for k in range(len(arr)):
                s = bytes(arr[k]).decode()
                assert (db[k], e[k]) == oracle.mfe(s)
                o = oracle.pf(s)
                assert o["centroid"] == r["centroid"][k] and abs(o["dG"] - r["dG"][k]) < 1e-9
                assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < 1e-9
    finally:
        emul.set_kernel_mode(0)
        emul.set_max_bp_span(0)
        oracle.set_max_bp_span(0)


def test_two_resident_models_switch_without_reload(emul, oracle):
    """sf_params_load keeps two models resident (a -t scan alternates between T and 37 C per chunk): going back to a set that is
    still in a slot must give that set's results again, a third set replaces the one not in use, and a max_bp_span set
    while the OTHER model was resident applies after the switch."""
    from par_util import par_text, synthetic_enthalpies
    base = params.default_params()
    p = params.parse_par_text(par_text(base.rec, synthetic_enthalpies(base.rec, 3)), source="synthetic.par")
    rng = np.random.default_rng(8)
    arr = random_seqs(rng, 4, 60)
    want = {}
    try:
        for t in (37, 25, 50):
            oracle.set_params(p.at_temperature(t))
            want[t] = (oracle.mfe_batch(arr), [oracle.pf(bytes(a).decode())["dG"] for a in arr])
        emul.load_params(p)
        for t in (37, 25, 37, 25, 50, 25, 37, 50):  # hits, a miss that evicts, then the evicted one again
            emul.set_temperature(t)
            assert (emul.mfe_batch(arr) == want[t][0]).all(), t
            assert np.allclose(emul.pf_batch(arr)["dG"], want[t][1], rtol=0, atol=1e-8), t
        # span set while 25 C is resident, then back to 37 C (still in its slot)
        emul.set_temperature(37); emul.set_temperature(25)
        emul.set_max_bp_span(20); oracle.set_max_bp_span(20)
        emul.set_temperature(37)
        oracle.set_params(p.at_temperature(37))
        assert (emul.mfe_batch(arr) == oracle.mfe_batch(arr)).all()
        emul.set_temperature(25)
        oracle.set_params(p.at_temperature(25))
        assert (emul.mfe_batch(arr) == oracle.mfe_batch(arr)).all()
    finally:
        emul.set_max_bp_span(0); oracle.set_max_bp_span(0)
        oracle.set_params(base)
        emul.load_params(base)


def test_poisoned_lds_slack_does_not_move_an_energy_emulated(emul, oracle, monkeypatch):
    """SCANFOLD_MFE_POISON (sf_mfe_fast.hip.h, PZ): before every fold each LDS byte the fold has not written itself holds
    -32768 / -28000 / 0 / 32767 / a mix.  The short-diagonal cell code over-reads rows no diagonal has written yet and the
    slack behind them; energies and tracebacks must equal the oracle under every pattern — on workgroups that fold several
    sequences in a row (the emulated device has few CUs), at one width per instantiation.  The GPU twin of this test is
    tests/test_gpu_paths.py::test_poisoned_lds_slack_does_not_move_an_energy."""
    emul.load_params(params.default_params())
    rng = np.random.default_rng(5)
    cases = [(W, random_seqs(rng, n, W)) for W, n in ((16, 12), (40, 10), (77, 8), (117, 6), (120, 6), (128, 5), (131, 3), (200, 2))]
    refs = [(oracle.mfe_batch(arr), [oracle.mfe(bytes(r).decode()) for r in arr[:2]]) for _, arr in cases]
    for pattern in ("1", "2", "3", "4", "5"):
        monkeypatch.setenv("SCANFOLD_MFE_POISON", pattern)
        for (W, arr), (ref, traces) in zip(cases, refs):
            assert (emul.mfe_batch(arr) == ref).all(), (pattern, W)
            e, db = emul.mfe_trace_batch(arr[:2])
            assert [(db[k], e[k]) for k in range(2)] == traces, (pattern, W)
    monkeypatch.delenv("SCANFOLD_MFE_POISON")
    for (W, arr), (ref, _) in zip(cases, refs):
        assert (emul.mfe_batch(arr) == ref).all(), W


def test_device_table_partition_function_cooperative_sums(emul, oracle):
    """sf_pf_fast_kernel (120 < W <= 256; tables in device memory): the O(W) multiloop sums of a diagonal are split over the
    diagonal's idle threads and added up from LDS by the cell's owner (inside_sums / outside_sums).  Widths on both sides of
    the 128- / 256-thread instantiations (W = 121: 128 threads, one part per cell until d = 57; W = 129: 256 threads, two parts
    from the first diagonal on; W = 250: one part until d = 122): == oracle."""
    rng = np.random.default_rng(12)
    for W, n in ((121, 2), (129, 2), (200, 1), (250, 1)):
        arr = random_seqs(rng, n, W)
        r = emul.pf_batch(arr)
        for k in range(n):
            o = oracle.pf(bytes(arr[k]).decode())
            assert abs(o["dG"] - r["dG"][k]) < 1e-9 and o["centroid"] == r["centroid"][k], W
            assert abs(o["mean_bp_dist"] - r["mean_bp_dist"][k]) < 1e-9 and abs(o["centroid_dist"] - r["centroid_dist"][k]) < 1e-9, W

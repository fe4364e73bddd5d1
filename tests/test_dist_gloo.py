"""The N>1 path on CPU: window sharding + the single gather, world_size 2 over gloo."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from scanfold_amd import dist as sdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_everything_once():
    for n in (0, 1, 7, 23, 989, 29881):
        for world in (1, 2, 4, 8):
            spans = [sdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) <= sdist.shard_size(n, world)


def fake_records(lo, hi, W, r):
    idx = np.arange(lo, hi)
    e = (idx[:, None] * 1000 + np.arange(r + 1)[None, :]).astype(np.int32) * -1
    s = np.zeros((hi - lo, W + 1), dtype=np.uint8); s[:, :W] = ord("."); s[:, 0] = 40 + (idx % 50)
    c = np.zeros((hi - lo, W + 1), dtype=np.uint8); c[:, :W] = ord("."); c[:, 1] = 41 + (idx % 50)
    return e, s, c, idx * 0.5, idx * -0.25


def test_pack_unpack_roundtrip_numpy():
    W, r, n = 37, 10, 13
    e, s, c, d, g = fake_records(0, n, W, r)
    rec = sdist.pack_records(np, W, r, e, s, c, d, g, n + 3)
    assert rec.shape == (n + 3, sdist.record_layout(W, r)["size"]) and sdist.record_layout(W, r)["size"] % 8 == 0
    out = sdist.unpack_records(rec, W, r, n)
    assert (out["energies"] == e).all() and (out["structure"] == s).all() and (out["centroid"] == c).all()
    assert (out["ens_div"] == d).all() and (out["ens_dG"] == g).all()


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from scanfold_amd import dist as sdist
    from test_dist_gloo import fake_records
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, r, n_win = 24, 5, int(os.environ["N_WIN"])   # 11 windows over 2 ranks: ragged last shard; 1 or 5 over 2 or 4: empty shards
    def produce(lo, hi):
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in fake_records(lo, hi, W, r))
    out = sdist.scan_sharded(produce, n_win, W, r, rank, world, torch)
    e, s, c, d, g = fake_records(0, n_win, W, r)
    ok = (out["energies"] == e).all() and (out["structure"] == s).all() and (out["centroid"] == c).all() \\
        and (out["ens_div"] == d).all() and (out["ens_dG"] == g).all()
    print("RANK%%d_OK=%%s" %% (rank, bool(ok)))
    dist.destroy_process_group()
""")


def run_ranks(script, nproc, env=None, timeout=300):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                          capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})))


@pytest.mark.parametrize("nproc,n_win", [(2, 11), (2, 1), (4, 5)])
def test_gather_over_gloo_ragged_and_empty_shards(tmp_path, nproc, n_win):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    out = run_ranks(script, nproc, {"N_WIN": str(n_win)})
    assert out.returncode == 0, out.stderr[-2000:]
    for rank in range(nproc):
        assert "RANK%d_OK=True" % rank in out.stdout, out.stdout + out.stderr[-1000:]


ENGINE_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests", "emul"))
    import numpy as np, torch, torch.distributed as dist
    from scanfold_amd import dist as sdist, scan as scanmod, _lib
    from scanfold_amd._lib import Engine
    from emul_engine import EMUL_LIB
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    eng = Engine(device=0, lib_path=EMUL_LIB)      # the kernel sources compiled for the CPU (test only)
    seq = "".join("ACGU"[k] for k in np.random.default_rng(5).integers(0, 4, 75))
    W, step, r, seed = 30, 4, 3, 9
    starts = scanmod.window_starts(len(seq), W, step)
    def produce(lo, hi):
        res = eng.scan(seq, W, step, lo, hi - lo, r, _lib.SHUFFLE_DI, seed, raw=True)
        return tuple(torch.from_numpy(res[k]) for k in ("energies", "structure", "centroid", "ens_div", "ens_dG"))
    m = sdist.scan_sharded(produce, len(starts), W, r, rank, world, torch)
    rows = scanmod.rows_from_results(seq, starts, W, r, 37, m["energies"], m["structure"], m["centroid"], m["ens_div"])
    whole = scanmod.scan_record(seq, W, step, r, "di", 37, eng, seed=seed)
    print("RANK%%d_ROWS_EQUAL=%%s n=%%d" %% (rank, rows == whole, len(rows)))
    dist.destroy_process_group()
""")


def test_engine_scan_sharded_over_two_gloo_ranks_equals_single_process(tmp_path):
    """The real engine code (kernel sources compiled for the CPU, tests/emul) behind dist.scan_sharded on two gloo ranks:
    the merged TSV rows equal the single-process scan — a window's shuffles depend on its absolute index only."""
    from emul_engine import build
    build()
    script = tmp_path / "engine_worker.py"
    script.write_text(ENGINE_WORKER % {"root": ROOT})
    out = run_ranks(script, 2, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RANK0_ROWS_EQUAL=True n=12" in out.stdout and "RANK1_ROWS_EQUAL=True n=12" in out.stdout, \
        out.stdout + out.stderr[-1500:]


def _write_fasta(path, seq, name="rec1"):
    with open(path, "w") as f:
        f.write(">%s test record\n" % name)
        for k in range(0, len(seq), 60):
            f.write(seq[k:k + 60] + "\n")


def _run_cli(args, env, timeout=900):
    return subprocess.run([sys.executable, "-m", "scanfold_amd.scan"] + args, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout, env=dict(os.environ, OMP_NUM_THREADS="1", **env))


def test_cli_gpus_2_over_gloo_writes_the_single_process_tsv(tmp_path):
    """`python -m scanfold_amd.scan --gpus 2` end to end on the CPU build of the engine (tests/emul), gloo gather: every
    rank scans and FORMATS its own window range, one gather of row slots, rank 0 writes — the file equals the one-process
    file byte for byte, also with a constraint line (-c) and an odd number of windows (ragged last shard)."""
    from emul_engine import build, EMUL_LIB
    build()
    seq = "".join("ACGT"[k] for k in np.random.default_rng(8).integers(0, 4, 83))
    fa = tmp_path / "in.fa"
    _write_fasta(fa, seq)
    cons = tmp_path / "cons.txt"
    cons.write_text(">rec1\n" + seq + "\n" + "".join("x" if k % 11 == 0 else "." for k in range(len(seq))) + "\n")
    env = {"SCANFOLD_LIB_PATH": EMUL_LIB, "SCANFOLD_DEVICE": "0", "SCANFOLD_DIST_BACKEND": "gloo"}
    for extra in ([], ["-c", str(cons)]):
        base = ["-i", str(fa), "-w", "30", "-s", "6", "-r", "3", "-type", "di", "--seed", "4"] + extra
        one, two = tmp_path / "one.tsv", tmp_path / "two.tsv"
        a = _run_cli(base + ["-o", str(one)], env)
        assert a.returncode == 0, a.stderr[-2000:]
        b = _run_cli(base + ["--gpus", "2", "-o", str(two)], env)
        assert b.returncode == 0, b.stderr[-2000:]
        assert one.read_bytes() == two.read_bytes() and one.read_text().count("\n") == 1 + 9, extra


ROWS_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    from scanfold_amd import dist as sdist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_win, W = int(os.environ["N_WIN"]), 20
    allrows = ["%%d\\t%%d\\t37\\t-%%d.5\\t%%s\\n" %% (k + 1, k + W, k, "ACGU" * 5) for k in range(n_win)]
    lo, hi = sdist.shard_range(n_win, rank, world)
    got = sdist.gather_rows(allrows[lo:hi], n_win, rank, world, W, device=None, want=(rank == 0))
    ok = (got == allrows) if rank == 0 else (got is None)
    print("RANK%%d_ROWS_OK=%%s" %% (rank, bool(ok)))
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("nproc,n_win", [(2, 7), (2, 1), (4, 2)])
def test_row_gather_over_gloo_ragged_and_empty_shards(tmp_path, nproc, n_win):
    """dist.gather_rows (the command line's one collective): ragged last shard, ranks with no window at all; only the
    writing rank unpacks."""
    script = tmp_path / "rows_worker.py"
    script.write_text(ROWS_WORKER % {"root": ROOT})
    out = run_ranks(script, nproc, {"N_WIN": str(n_win)})
    assert out.returncode == 0, out.stderr[-2000:]
    for rank in range(nproc):
        assert "RANK%d_ROWS_OK=True" % rank in out.stdout, out.stdout + out.stderr[-1000:]


def test_cli_gpus_2_rank_local_constraint_error_fails_on_every_rank_instead_of_hanging(tmp_path):
    """A constraint whose brackets are unbalanced only in windows of the SECOND rank's shard: that rank raises before the
    gather.  Every rank must learn it (one all-reduce of a status flag) and end with an error — the other rank must not sit
    in all_gather_into_tensor until the collective times out."""
    from emul_engine import build, EMUL_LIB
    build()
    seq = "".join("ACGU"[k] for k in np.random.default_rng(8).integers(0, 4, 83))
    fa = tmp_path / "in.fa"
    _write_fasta(fa, seq)
    line = ["."] * len(seq)
    line[70], line[80] = "(", ")"   # complete in the last windows only ...
    line[60] = "("                  # ... and an opening bracket that never closes inside the windows that hold it
    cons = tmp_path / "cons.txt"
    cons.write_text(">rec1\n" + seq + "\n" + "".join(line) + "\n")
    env = {"SCANFOLD_LIB_PATH": EMUL_LIB, "SCANFOLD_DEVICE": "0", "SCANFOLD_DIST_BACKEND": "gloo"}
    b = _run_cli(["-i", str(fa), "-w", "30", "-s", "6", "-r", "2", "-type", "di", "--seed", "4", "-c", str(cons), "--gpus", "2",
                  "-o", str(tmp_path / "two.tsv")], env, timeout=240)
    assert b.returncode != 0
    assert "unbalanced" in b.stderr and "another rank failed" in b.stderr, b.stderr[-3000:]


def test_engine_chunks_joins_its_helper_thread_before_the_caller_touches_the_library_again():
    """scan._engine_chunks: when the consumer stops early (formatting raised, generator closed), the helper thread — the
    only one allowed inside the non-reentrant library while it lives — must have left work() before control returns."""
    import threading
    import time
    from scanfold_amd import scan as scanmod
    inside, log = threading.Event(), []

    def work(w0, nw):
        inside.set()
        time.sleep(0.15)
        log.append(("done", w0))
        inside.clear()
        return {"ens_div": [0] * nw}

    gen = scanmod._engine_chunks(work, 0, 50, chunk=5, threaded=True)
    first = next(gen)
    assert first[0] == 0
    gen.close()  # what a raising consumer does to the generator
    assert not inside.is_set()
    n_done = len(log)
    time.sleep(0.4)
    assert len(log) == n_done and n_done <= 4  # nothing ran after close(); the remaining chunks were never started
    with pytest.raises(RuntimeError, match="boom"):
        for w0, _ in scanmod._engine_chunks(work, 0, 50, chunk=5, threaded=True):
            raise RuntimeError("boom")
    assert not inside.is_set()

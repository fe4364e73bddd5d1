"""bench.py's launch contract, as far as a machine without GPUs can show it.

`python bench.py --gpus N` must work when the driver calls it plainly: it starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a child, relays the JSON line and the exit code.  Here there is no GPU, so
every rank stops at device selection with a clear message — what matters is that N ranks were spawned and that
the failure is that one, not a usage error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None):
    e = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                          timeout=600, env=e)


def test_gpus_2_spawns_two_ranks_and_fails_only_at_device_selection():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs present: the real run is the driver's job")
    out = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert out.returncode != 0
    assert "rank 0: --gpus 2 needs 2 GPUs" in out.stderr and "rank 1: --gpus 2 needs 2 GPUs" in out.stderr, out.stderr[-3000:]
    assert "GPU(s) visible" in out.stderr
    assert '"metric"' not in out.stdout


def test_world_size_mismatch_is_reported():
    out = run_bench("--gpus", "1", env={"RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr


def test_calibrated_unit_fractions_from_the_committed_rates():
    """bench.py's roofline.secondary: instruction counts x the instruction costs measured on MI355X (profiles/r05/
    mfe_issue_rates.json: tools/micro/issue_rates.hip, valu_classes.hip; the kernel's vector mix from its round-5 ISA) / the launch
    time.  The committed cfg3 counters must reproduce the fractions the committed bench line carries, and the rates file must say
    what it was measured on."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    rates = json.load(open(os.path.join(ROOT, "profiles", "r05", "mfe_issue_rates.json")))
    assert 0.9 < rates["valu_full_rate_ns"] < 1.3 and 1.6 < rates["valu_half_rate_ns"] < 2.2
    assert "v_pk_min_i16" in rates["valu_half_rate_opcodes"] and "v_add_u32" in rates["valu_full_rate_opcodes"]
    # round 5 moved a part of the kernel onto full-rate 16-bit instructions: its mix got cheaper than round 4's 1.715 ns
    assert rates["valu_full_rate_ns"] < rates["valu_ns_per_inst_per_simd"]["120"] < 1.715
    line = json.load(open(os.path.join(ROOT, "profiles", "r05", "final_bench.json")))
    sec = line["roofline"]["secondary"]
    got = bench.calibrated_unit_fractions(sec, 120, line["roofline"]["folds_per_launch"], line["roofline"]["avg_launch_ms"])
    assert abs(got["valu_issue_frac"] - sec["valu_issue_frac"]) < 1e-9 and abs(got["lds_frac"] - sec["lds_frac"]) < 1e-9
    assert 0.5 < got["valu_issue_frac"] < 1.0 and got["binding_unit"] == "valu"
    # the counters file bench.py falls back to (--no-live-counters) describes the same launch
    cnt = json.load(open(os.path.join(ROOT, "profiles", "r05", "mfe_counters.json")))
    assert cnt["folds"] == 3017981 and cnt["launches"] == 1
    assert abs(cnt["secondary"]["valu_insts_per_fold"] / sec["valu_insts_per_fold"] - 1) < 0.02
    # (two sets of counter passes of the same scan; the traffic is capacity evictions of the scratch tables from the L2s — a few
    # hundred bytes per fold, +-20 % between passes: 2.4 and 2.8 GB per launch in round 5, 1.7 GB in round 4, 16.5 GB before the
    # tables of one XCD's workgroups were made adjacent)
    assert abs(cnt["hbm_bytes_per_launch"] / line["roofline"]["traffic"] - 1) < 0.25
    assert cnt["hbm_bytes_per_launch"] < 3.5e9 and line["roofline"]["traffic"] < 3.5e9
    # the headline the round's documents quote
    assert line["value"] > 43000 and line["verified_mismatches"] == 0 and line["e2e"]["windows_per_s"] > 42000


def test_chunk_schedule_covers_the_range_and_tapers():
    """scan._chunk_bounds: the engine calls of a scan cover [lo, hi) without gaps; the default schedule ends in a small chunk
    (what the pipeline cannot overlap is the host's work on the last one), an explicit size gives equal chunks."""
    sys.path.insert(0, ROOT)
    from scanfold_amd import scan
    for n in (0, 1, 255, 4096, 4097, 5000, 29881, 100000):
        b = scan._chunk_bounds(10, 10 + n)
        assert sum(nw for _, nw in b) == n and all(nw > 0 for _, nw in b)
        assert all(b[k][0] + b[k][1] == b[k + 1][0] for k in range(len(b) - 1)) and (not b or b[0][0] == 10)
        if n > scan.CHUNK_WINDOWS:
            assert b[-1][1] == 512 and max(nw for _, nw in b) <= 2 * scan.CHUNK_WINDOWS
    assert [nw for _, nw in scan._chunk_bounds(0, 29881)] == [7343, 7343, 7343, 7340, 512]
    assert scan._chunk_bounds(0, 10000, chunk=3000) == [(0, 3000), (3000, 3000), (6000, 3000), (9000, 1000)]


def test_chunk_size_must_be_positive(monkeypatch):
    """A zero or negative explicit chunk (argument or SCANFOLD_CHUNK_WINDOWS) is refused instead of producing range(lo, hi, 0)
    or an empty schedule (a scan that silently writes no rows)."""
    import pytest
    from scanfold_amd import scan
    with pytest.raises(ValueError):
        scan._chunk_bounds(0, 100, chunk=-5)
    monkeypatch.setenv("SCANFOLD_CHUNK_WINDOWS", "0")
    with pytest.raises(ValueError):
        scan._chunk_bounds(0, 100)
    monkeypatch.setenv("SCANFOLD_CHUNK_WINDOWS", "-1")
    with pytest.raises(ValueError):
        scan._chunk_bounds(0, 100)


def test_gather_rows_takes_a_prepacked_shard_of_the_right_shape_only():
    import numpy as np
    import pytest
    from scanfold_amd import dist as sdist
    rows = ["1\t120\tx\n", "2\t121\ty\n"]
    assert sdist.gather_rows(rows, 2, 0, 1, 120) == rows  # one rank: nothing is packed or exchanged
    # world 2 without a process group cannot gather, but the shape check comes first
    bad = np.zeros((3, 7), dtype=np.uint8)
    with pytest.raises(ValueError, match="prepacked"):
        sdist.gather_rows(rows[:1], 2, 0, 2, 120, prepacked=bad)

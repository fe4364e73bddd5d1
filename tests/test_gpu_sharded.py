"""The sharded command line on the real GPU: two ranks — both on GPU 0, gathered over gloo, because RCCL refuses two
ranks on one device and the test box has one — started through torch.distributed.run as a child process by
`python -m scanfold_amd.scan --gpus 2`.  Matches north_star's "shard across the GPUs ... single gather at the end";
the exchange itself (dist.gather_rows: one all_gather_into_tensor of row slots) is backend-agnostic."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run_cli(args, env, timeout=900):
    return subprocess.run([sys.executable, "-m", "scanfold_amd.scan"] + args, cwd=ROOT, capture_output=True, text=True,
                          timeout=timeout, env=dict(os.environ, **env))


def test_cli_two_ranks_on_gpu0_write_the_single_process_tsv_cfg2(tmp_path, gpu_engine):
    """BASELINE config 2 (10 kb, W = 120, step = 10, 30 di-shuffles): the TSV of `--gpus 2` equals the one-process file
    byte for byte — a window's shuffles depend on (seed, absolute window index) only, every rank formats its own rows."""
    seq = "".join("ACGU"[k] for k in np.random.default_rng(2).integers(0, 4, 10000))
    fa = tmp_path / "cfg2.fa"
    with open(fa, "w") as f:
        f.write(">cfg2 synthetic\n")
        for k in range(0, len(seq), 80):
            f.write(seq[k:k + 80] + "\n")
    base = ["-i", str(fa), "-w", "120", "-s", "10", "-r", "30", "-type", "di", "--seed", "11", "--timing"]
    env = {"SCANFOLD_DEVICE": "0", "SCANFOLD_DIST_BACKEND": "gloo"}
    one, two = tmp_path / "one.tsv", tmp_path / "two.tsv"
    a = _run_cli(base + ["-o", str(one)], env)
    assert a.returncode == 0, a.stderr[-3000:]
    b = _run_cli(base + ["--gpus", "2", "-o", str(two)], env)
    assert b.returncode == 0, b.stderr[-3000:]
    assert one.read_bytes() == two.read_bytes()
    assert one.read_text().count("\n") == 1 + 989
    lines = [ln for ln in b.stderr.splitlines() if "timing rank" in ln]
    assert len(lines) == 2 and "windows=495" in lines[0] + lines[1] and "windows=494" in lines[0] + lines[1], b.stderr[-2000:]
    # and the first rows carry what the engine in this process computes for the same windows
    res = gpu_engine.scan(seq, 120, 10, 0, 4, 30, 1, 11)
    first = two.read_text().splitlines()[1].split("\t")
    assert first[0] == "1" and first[1] == "120" and first[8] == res["structure"][0]
